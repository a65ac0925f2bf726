"""A/B of the purification step with one of the engine's routing flags off and on, alternating inside one process:
python scripts/ab_purify_attr.py _fuse_head_post [image size] [batch] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dvae_amd import _lib as L  # noqa: E402
from dvae_amd.engine import DVAEConfig, DVAEEngine  # noqa: E402

attr = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
Bp = int(sys.argv[3]) if len(sys.argv) > 3 else {32: 2048, 64: 1024, 224: 128}[S]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
cfg = DVAEConfig(**bench.model_kwargs(S))
torch.manual_seed(0)
eng = DVAEEngine(cfg, Bp, "cuda", L.BF16, seed=0)
bench.init_params(eng)
assert getattr(eng, attr), attr + " is off for this configuration"
n = 4 * Bp
data = torch.randint(0, 256, (n, S, S, 3), dtype=torch.uint8, device="cuda")
labels = (torch.arange(n, device="cuda") % cfg.num_classes).long()
out = torch.empty_like(data)
noise = torch.empty(n, 3, S, S, device="cuda")
for rep in range(3):
    for v in (False, True):
        setattr(eng, attr, v)
        for i in range(5):
            eng.purify_batch(data, out, noise, labels, (i % 4) * Bp, Bp, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.purify_batch(data, out, noise, labels, (i % 4) * Bp, Bp, 1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"S={S} {attr}={v}: {dt * 1e3:.3f} ms/step  {Bp / dt / 1e3:.2f} k images/s", flush=True)
