import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import torch.nn.functional as F
from oracle import dvae_oracle as O
import dvae_amd
from dvae_amd import _lib as L
from dvae_amd.engine import DVAEConfig, DVAEEngine
from util_gpu import rel_err, from_act

ocfg = O.DVAEConfig(); cfg = DVAEConfig()
eng = DVAEEngine(cfg, 4, "cuda", L.F32)
p = O.recipe_params(ocfg); eng.load_params(p)
x, y, eps = O.make_inputs(ocfg, 4)
leaf = {k: (v.clone().requires_grad_(True) if not O.is_buffer(k) else v.clone()) for k, v in p.items()}
# manual oracle forward with retained intermediates
inter = {}
def bn(prefix, t):
    return F.batch_norm(t, leaf[prefix+".running_mean"].clone(), leaf[prefix+".running_var"].clone(), leaf[prefix+".weight"], leaf[prefix+".bias"], True, 0.1, 1e-5)
a = F.relu(bn("bn1", F.conv2d(x, leaf["conv1.weight"], None, 1, 1))); a.retain_grad(); inter["bn1.out"] = a
for blk in ["layer1.0", "layer1.1", "layer2.0"]:
    s = 2 if blk == "layer2.0" else 1
    c1 = F.conv2d(a, leaf[blk+".conv1.weight"], None, s, 1); c1.retain_grad(); inter[blk+".conv1.out"] = c1
    t = F.relu(bn(blk+".bn1", c1)); t.retain_grad(); inter[blk+".bn1.out"] = t
    c2 = F.conv2d(t, leaf[blk+".conv2.weight"], None, 1, 1); c2.retain_grad(); inter[blk+".conv2.out"] = c2
    o = bn(blk+".bn2", c2)
    if blk == "layer2.0":
        cs = F.conv2d(a, leaf[blk+".shortcut.0.weight"], None, s, 0); cs.retain_grad(); inter[blk+".shortcut.0.out"] = cs
        sc = bn(blk+".shortcut.1", cs)
    else:
        sc = a
    a = F.relu(o + sc); a.retain_grad(); inter[blk+".out"] = a
stem = a
mu = F.conv2d(stem, leaf["fc_mu.weight"], leaf["fc_mu.bias"]); lv = F.conv2d(stem, leaf["fc_var.weight"], leaf["fc_var.bias"])
z1 = eps * torch.exp(0.5*lv) + mu
z = F.conv2d(z1, leaf["post_conv.weight"], leaf["post_conv.bias"]); z.retain_grad(); inter["z"] = z
rec = O._decoder(leaf, "decoder", z, True)
emb = leaf["emb_p.weight"][y].reshape(stem.shape)
pert = O._decoder(leaf, "decoder_p", emb + z, False)
Ls = O.losses(dict(rec=rec, perturbation=pert, mu=mu, log_var=lv), x, 0, 1.0, 0.01)
Ls["loss"].backward()

x0 = eng.input_from_nchw(x.cuda()); e = eng.eps_from_nchw(eps.cuda())
eng.forward(x0, y.cuda(), e, 4, True); eng.loss_sums(); eng.loss_gate(0, 1.0, 0.01); eng.backward()
def cmp(name, gname, c):
    ref = inter[name].grad
    got = from_act(eng._bufs[gname][:4], c)
    print(f"{name:28s} grad rel={rel_err(got, ref):.2e}  fwd rel={rel_err(from_act(eng._bufs[name][:4], c), inter[name].detach()):.2e}")
cmp("layer2.0.out", "g.stem_out", 128)
cmp("layer2.0.conv2.out", "g.layer2.0.conv2.out", 128)
cmp("layer2.0.bn1.out", "g.layer2.0.bn1.out", 128)
cmp("layer2.0.conv1.out", "g.layer2.0.conv1.out", 128)
cmp("layer2.0.shortcut.0.out", "g.layer2.0.shortcut.0.out", 128)
cmp("layer1.1.out", "g.layer2.0.in", 64)
cmp("layer1.1.conv2.out", "g.layer1.1.conv2.out", 64)
cmp("layer1.0.out", "g.layer1.1.in", 64)
cmp("bn1.out", "g.layer1.0.in", 64)
print("---- isolate bn2 bwd of layer2.0 using engine buffers")
B = eng._bufs
xx = B["layer2.0.conv2.out"][:4].clone().reshape(-1, 128).requires_grad_(True)
ss = B["layer2.0.shortcut.1.out"][:4].clone().reshape(-1, 128).requires_grad_(True)
ga = eng.param["layer2.0.bn2.weight"].clone().requires_grad_(True); be = eng.param["layer2.0.bn2.bias"].clone().requires_grad_(True)
mean = xx.mean(0, keepdim=True); var = xx.var(0, unbiased=False, keepdim=True)
yy = F.relu((xx - mean) / torch.sqrt(var + 1e-5) * ga + be + ss)
print("y recompute vs engine out", rel_err(yy.detach(), B["layer2.0.out"][:4].reshape(-1, 128)))
dy = B["g.stem_out"][:4].reshape(-1, 128)
yy.backward(dy)
print("dx   ", rel_err(B["g.layer2.0.conv2.out"][:4].reshape(-1,128), xx.grad))
print("dres ", rel_err(B["g.layer2.0.res"][:4].reshape(-1,128), ss.grad))
print("dgam ", rel_err(eng.grad["layer2.0.bn2.weight"], ga.grad), "dbeta", rel_err(eng.grad["layer2.0.bn2.bias"], be.grad))
m, i = eng.bn_save["layer2.0.bn2"]
print("mean ", rel_err(m, mean.detach().reshape(-1)), "invstd", rel_err(i, (1/torch.sqrt(var+1e-5)).detach().reshape(-1)))
