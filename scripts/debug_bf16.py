import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from oracle import dvae_oracle as O
import dvae_amd
from dvae_amd import _lib as L
from dvae_amd.engine import DVAEConfig, DVAEEngine
from util_gpu import rel_err, l2_err, from_act
ocfg, cfg = O.DVAEConfig(), DVAEConfig()
p = O.recipe_params(ocfg)
for B in (8, 64):
    eng = DVAEEngine(cfg, B, "cuda", L.BF16); eng.load_params(p)
    x, y, eps = O.make_inputs(ocfg, B)
    info, grads, _ = O.train_step({k: v.clone() for k, v in p.items()}, ocfg, x, y, eps, {}, 11, 1)
    x0 = eng.input_from_nchw(x.cuda()); e = eng.eps_from_nchw(eps.cuda())
    eng.forward(x0, y.cuda(), e, B, True); eng.loss_sums(); eng.loss_gate(11, 3.0, 0.01); eng.backward()
    print("B", B, {k: (float(eng.scal[L.SCAL[k]]), info[k]) for k in ("loss", "kld", "loss_p")})
    print("  ", " ".join(f"{k.replace('.weight','.w').replace('.bias','.b')}={l2_err(eng.grad[k].cpu(), g):.3f}" for k, g in grads.items() if not k.endswith('1.1.bias') and not k.endswith('2.0.bias')))
    gn = torch.cat([eng.grad[k].flatten().cpu() for k in grads]); go = torch.cat([g.flatten() for g in grads.values()])
    print("   whole-bucket L2 err", float((gn-go).norm()/go.norm()), "cos", float((gn*go).sum()/gn.norm()/go.norm()))
# purification diff histogram
for stage in (0, 1):
    outs = {}
    N = 512
    rng = np.random.default_rng(1)
    data = torch.from_numpy(rng.integers(0, 256, (N, 32, 32, 3), dtype=np.uint8)).cuda()
    labels = (torch.arange(N, device="cuda") % 10).long()
    for name, dt in (("f32", L.F32), ("bf16", L.BF16)):
        eng = DVAEEngine(cfg, 256, "cuda", dt); eng.load_params(p); eng.seed = 5
        out = torch.zeros_like(data); noise = torch.zeros(N, 3, 32, 32, device="cuda")
        for lo in range(0, N, 256):
            eng.purify_batch(data, out, noise, labels, lo, 256, stage)
        outs[name] = (out, noise)
    d = (outs["f32"][0].int() - outs["bf16"][0].int()).abs().flatten()
    print("stage", stage, "hist", torch.bincount(d, minlength=12).tolist(), "noise l2", l2_err(outs["bf16"][1], outs["f32"][1]))
