"""A/B of the two 3x3 64-input-channel conv kernels INSIDE the purification step (the diagnostic switch
dvae_set_option("conv_wreg", v)), alternating in one process: images/s of the timed region and the HIP-event time of the
probed launch.  The card is power-limited under this load, so a kernel compared alone says little about the step."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402
from dvae_amd.engine import DVAEConfig, DVAEEngine  # noqa: E402
import bench  # noqa: E402

a = argparse.Namespace(warmup=3, stage=1)
Bp, S = 1024, 32
cfg = DVAEConfig()
eng = DVAEEngine(cfg, Bp, "cuda", L.BF16, seed=0)
bench.init_params(eng)
g = torch.Generator(device="cuda").manual_seed(0)
n = 20 * Bp
data = torch.randint(0, 256, (n, S, S, 3), generator=g, dtype=torch.uint8, device="cuda")
labels = (torch.arange(n, device="cuda") % cfg.num_classes).long()
out = torch.empty_like(data)
noise = torch.empty(n, 3, S, S, device="cuda")
lib = L.lib()
for rep in range(int(os.environ.get("AB_REPS", "4"))):
    for v in (1, 0):
        L.check(lib.dvae_set_option(b"conv_wreg", v))
        ips, ms, probe = bench.purify_region(eng, a, 1, data, out, noise, labels, Bp, 20, bench.CONV_PROBE)
        print(f"rep {rep} conv_wreg={v}: {ips / 1e3:7.1f} k images/s, {ms:.3f} ms/step, probe {sum(probe) / len(probe) * 1e3:.1f} us", flush=True)
L.check(lib.dvae_set_option(b"conv_wreg", 1))
