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
def run(dt):
    leaf = {k: (v.to(dt).clone().requires_grad_(True) if not O.is_buffer(k) else v.to(dt).clone()) for k, v in p.items()}
    inter = {}
    stem = O.forward_stem(leaf, ocfg, x.to(dt), True); stem.retain_grad(); inter["stem"] = stem
    mu = F.conv2d(stem, leaf["fc_mu.weight"], leaf["fc_mu.bias"]); lv = F.conv2d(stem, leaf["fc_var.weight"], leaf["fc_var.bias"])
    mu.retain_grad(); lv.retain_grad(); inter["mu"] = mu; inter["lv"] = lv
    z1 = eps.to(dt) * torch.exp(0.5*lv) + mu; z1.retain_grad(); inter["z1"] = z1
    z = F.conv2d(z1, leaf["post_conv.weight"], leaf["post_conv.bias"]); z.retain_grad(); inter["z"] = z
    rec = O._decoder(leaf, "decoder", z, True)
    emb = leaf["emb_p.weight"][y].reshape(stem.shape)
    pert = O._decoder(leaf, "decoder_p", emb + z, False)
    Ls = O.losses(dict(rec=rec, perturbation=pert, mu=mu, log_var=lv), x.to(dt), 0, 1.0, 0.01)
    Ls["loss"].backward()
    return {k: v.grad for k, v in inter.items()}
g32 = run(torch.float32); g64 = run(torch.float64)
x0 = eng.input_from_nchw(x.cuda()); e = eng.eps_from_nchw(eps.cuda())
eng.forward(x0, y.cuda(), e, 4, True); eng.loss_sums(); eng.loss_gate(0, 1.0, 0.01); eng.backward()
def l2(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a-b).norm()/b.norm())
B = eng._bufs
got = {"stem": from_act(B["g.stem_out"][:4], 128), "z": from_act(B["g.dec_p.in"][:4], 128), "z1": from_act(B["g.z1"][:4], 256),
       "mu": from_act(B["g.mulv"][:4, :, :, :256], 256), "lv": from_act(B["g.mulv"][:4, :, :, 256:], 256)}
for k in ["z", "z1", "mu", "lv", "stem"]:
    print(f"{k:6s} gpu: max-rel={rel_err(got[k], g64[k]):.2e} l2-rel={l2(got[k], g64[k]):.2e} | cpu32: max-rel={rel_err(g32[k], g64[k]):.2e} l2-rel={l2(g32[k], g64[k]):.2e}  | max={float(g64[k].abs().max()):.3e} med={float(g64[k].abs().median()):.3e}")
