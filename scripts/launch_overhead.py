"""Fixed cost of a launch of the weights-in-registers 3x3 conv against its per-tile cost: the same 64 -> 64 conv at 32 x 32
over batches from 256 to 8192 images (1 to 32 images = 4 to 128 tiles per workgroup), each timed over a sustained run of
back-to-back launches.  A straight line through (tiles per workgroup, us per launch) gives both."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402

lib = L.lib()
st = L.stream()
hin, c = int(os.environ.get("HIN", "32")), 64
g = torch.Generator(device="cuda").manual_seed(0)
w = (torch.randn(c, c, 3, 3, device="cuda", generator=g) * 0.05).bfloat16().float()
scale = torch.rand(c, device="cuda", generator=g) + 0.5
shift = torch.randn(c, device="cuda", generator=g) * 0.2
pts = []
for B in [int(v) for v in (sys.argv[1:] or [256, 512, 1024, 2048, 4096, 8192])]:
    d = L.ConvDesc(B, hin, hin, c, c, 3, 3, 1, 1, 0)
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    x = torch.relu(torch.randn(B, hin, hin, c, device="cuda", generator=g)).bfloat16()
    y = torch.empty_like(x)
    ep = L.Epilogue(scale.data_ptr(), shift.data_ptr(), None, 1, 0)
    fn = lambda: L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))  # noqa: E731
    n = max(200, int(0.4 / (B * 75e-9)))
    for _ in range(n // 2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tiles = B * (hin // 8) * (hin // 32) / 256
    pts.append((tiles, us))
    print(f"B={B:5d}: {tiles:6.1f} tiles per workgroup, {us:8.1f} us per launch, {2 * 64 * 64 * 9 * hin * hin * B / us / 1e9:.3f} PFLOP/s", flush=True)
(t0, u0), (t1, u1) = pts[0], pts[-1]
per = (u1 - u0) / (t1 - t0)
print(f"per tile {per:.3f} us, fixed {u0 - per * t0:.1f} us per launch")
