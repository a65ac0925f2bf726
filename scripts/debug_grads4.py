import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import torch.nn.functional as F
from oracle import dvae_oracle as O
import dvae_amd
from dvae_amd import _lib as L
from dvae_amd.engine import DVAEConfig, DVAEEngine
from util_gpu import rel_err, from_act, to_act

ocfg = O.DVAEConfig(); cfg = DVAEConfig()
eng = DVAEEngine(cfg, 4, "cuda", L.F32)
p = O.recipe_params(ocfg); eng.load_params(p)
x, y, eps = O.make_inputs(ocfg, 4)
leaf = {k: (v.clone().requires_grad_(True) if not O.is_buffer(k) else v.clone()) for k, v in p.items()}
def bn(prefix, t):
    return F.batch_norm(t, leaf[prefix+".running_mean"].clone(), leaf[prefix+".running_var"].clone(), leaf[prefix+".weight"], leaf[prefix+".bias"], True, 0.1, 1e-5)
a = F.relu(bn("bn1", F.conv2d(x, leaf["conv1.weight"], None, 1, 1)))
for blk in ["layer1.0", "layer1.1", "layer2.0"]:
    s = 2 if blk == "layer2.0" else 1
    c1 = F.conv2d(a, leaf[blk+".conv1.weight"], None, s, 1)
    t = F.relu(bn(blk+".bn1", c1))
    c2 = F.conv2d(t, leaf[blk+".conv2.weight"], None, 1, 1); c2.retain_grad()
    o = bn(blk+".bn2", c2)
    if blk == "layer2.0":
        cs = F.conv2d(a, leaf[blk+".shortcut.0.weight"], None, s, 0)
        sc = bn(blk+".shortcut.1", cs); sc.retain_grad()
    else:
        sc = a
    a = F.relu(o + sc); a.retain_grad()
stem = a
mu = F.conv2d(stem, leaf["fc_mu.weight"], leaf["fc_mu.bias"]); lv = F.conv2d(stem, leaf["fc_var.weight"], leaf["fc_var.bias"])
z1 = eps * torch.exp(0.5*lv) + mu
z = F.conv2d(z1, leaf["post_conv.weight"], leaf["post_conv.bias"])
rec = O._decoder(leaf, "decoder", z, True)
emb = leaf["emb_p.weight"][y].reshape(stem.shape)
pert = O._decoder(leaf, "decoder_p", emb + z, False)
Ls = O.losses(dict(rec=rec, perturbation=pert, mu=mu, log_var=lv), x, 0, 1.0, 0.01)
Ls["loss"].backward()

x0 = eng.input_from_nchw(x.cuda()); e = eng.eps_from_nchw(eps.cuda())
eng.forward(x0, y.cuda(), e, 4, True); eng.loss_sums(); eng.loss_gate(0, 1.0, 0.01); eng.backward()
B = eng._bufs
print("engine vs oracle  d_c2:", rel_err(from_act(B["g.layer2.0.conv2.out"][:4], 128), c2.grad))
print("engine vs oracle  d_sc(gm):", rel_err(from_act(B["g.layer2.0.res"][:4], 128), sc.grad))
print("engine vs oracle  sc fwd:", rel_err(from_act(B["layer2.0.shortcut.1.out"][:4], 128), sc.detach()))
# feed ORACLE tensors into the kernel
lib = L.lib()
dy = to_act(stem.grad); yy = to_act(stem.detach()); xx = to_act(c2.detach())
dx = torch.empty_like(xx); dres = torch.empty_like(xx)
m, i = eng.bn_save["layer2.0.bn2"]
dg = torch.empty(128, device="cuda"); db = torch.empty(128, device="cuda")
L.check(lib.dvae_norm_bwd(dy.data_ptr(), yy.data_ptr(), xx.data_ptr(), dx.data_ptr(), dres.data_ptr(), eng.param["layer2.0.bn2.weight"].data_ptr(),
        m.data_ptr(), i.data_ptr(), dg.data_ptr(), db.data_ptr(), eng.norm_ws.data_ptr(), eng.norm_ws.numel()*4, 1, 1024, 128, 1, L.F32, L.stream()))
print("kernel(oracle inputs) vs oracle d_c2:", rel_err(from_act(dx, 128), c2.grad))
print("kernel(oracle inputs) vs engine d_c2:", rel_err(dx, B["g.layer2.0.conv2.out"][:4]))
print("dy oracle vs engine g.stem_out (act layout):", rel_err(dy, B["g.stem_out"][:4]), " y:", rel_err(yy, B["layer2.0.out"][:4]), " x:", rel_err(xx, B["layer2.0.conv2.out"][:4]))
