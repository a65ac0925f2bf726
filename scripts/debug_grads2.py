import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from oracle import dvae_oracle as O
import dvae_amd
from dvae_amd import _lib as L
from dvae_amd.engine import DVAEConfig, DVAEEngine
from util_gpu import rel_err

ocfg = O.DVAEConfig(); cfg = DVAEConfig()
eng = DVAEEngine(cfg, 4, "cuda", L.F32)
p = O.recipe_params(ocfg); eng.load_params(p)
x, y, eps = O.make_inputs(ocfg, 4)
_, g32, _ = O.train_step({k: v.clone() for k, v in p.items()}, ocfg, x, y, eps, {}, 0, 0)
_, g64, _ = O.train_step({k: v.double() for k, v in p.items()}, ocfg, x.double(), y, eps.double(), {}, 0, 0)
x0 = eng.input_from_nchw(x.cuda()); e = eng.eps_from_nchw(eps.cuda())
eng.forward(x0, y.cuda(), e, 4, True); eng.loss_sums(); eng.loss_gate(0, 1.0, 0.01); eng.backward()
for k in g32:
    print(f"{k:32s} gpu-vs-f64={rel_err(eng.grad[k].cpu(), g64[k]):.2e}  cpu32-vs-f64={rel_err(g32[k], g64[k]):.2e}")
