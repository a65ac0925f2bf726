"""A/B of the batch-128 train step with one of the library's diagnostic switches off and on, alternating inside one process:
python scripts/ab_train_opt.py conv_up [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dvae_amd import _lib as L  # noqa: E402
from dvae_amd.engine import DVAEConfig, DVAEEngine  # noqa: E402

opt = sys.argv[1].encode()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = DVAEConfig(**bench.model_kwargs(32))
torch.manual_seed(0)
eng = DVAEEngine(cfg, 2048, "cuda", L.BF16, seed=0)
bench.init_params(eng)
n, Bt = 50000, 128
data = torch.randint(0, 256, (n, 32, 32, 3), dtype=torch.uint8, device="cuda")
labels = (torch.arange(n, device="cuda") % cfg.num_classes).long()
eng.set_hyper(lr=5e-3, max_norm=5.0, grad_scale=1.0, steps_done=0)
idx = torch.randperm(n, device="cuda", dtype=torch.int64).to(torch.int32)


def train_step(i):
    lo = (i * Bt) % (n - Bt)
    sel = idx[lo:lo + Bt]
    x0 = eng.input_from_u8(data, 0, Bt, index=sel)
    eps = eng.eps_philox(Bt, index=sel, stream_id=100)
    eng.train_step(x0, labels[sel.long()], eps, Bt, 11, 3.0, 0.01, None, 1)


for rep in range(3):
    for v in (0, 1):
        L.check(L.lib().dvae_set_option(opt, v))
        for i in range(30):
            train_step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            train_step(30 + i)
        torch.cuda.synchronize()
        print(f"{opt.decode()}={v}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step", flush=True)
