"""Fixed cost per launch of the forward conv kernels of a purification step: every shape timed at 256 and at 1024 images
(sustained back-to-back launches); fixed = t(256) - (t(1024) - t(256)) / 3."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402

SHAPES = [  # name, cin, cout, k, stride, pad, transposed, hin
    ("conv1 (stem)", 3, 64, 3, 1, 1, 0, 32), ("layer1 64->64", 64, 64, 3, 1, 1, 0, 32), ("layer2.0.conv1 s2", 64, 128, 3, 2, 1, 0, 32),
    ("shortcut 1x1 s2", 64, 128, 1, 2, 0, 0, 32), ("layer2.0.conv2", 128, 128, 3, 1, 1, 0, 16), ("fc 128->512", 128, 512, 1, 1, 0, 0, 16),
    ("post_conv", 256, 128, 1, 1, 0, 0, 16), ("ConvT4 128->64", 128, 64, 4, 2, 1, 1, 16), ("ConvT3 64->64", 64, 64, 3, 1, 1, 1, 32),
    ("ConvT3 64->32", 64, 32, 3, 1, 1, 1, 32), ("ConvT3 32->3", 32, 3, 3, 1, 1, 1, 32),
]
lib = L.lib()
st = L.stream()
for name, cin, cout, k, stride, pad, tr, hin in SHAPES:
    ts = {}
    for B in (256, 1024):
        d = L.ConvDesc(B, hin, hin, cin, cout, k, k, stride, pad, tr)
        ho = (hin - 1) * stride - 2 * pad + k if tr else (hin + 2 * pad - k) // stride + 1
        w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device="cuda") * 0.05
        pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_conv_pack(w.data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
        x = torch.relu(torch.randn(B, hin, hin, L.act_channels(cin, L.BF16), device="cuda")).bfloat16()
        f32o = cout == 3
        y = torch.empty(B, ho, ho, L.act_channels(cout, L.F32 if f32o else L.BF16), device="cuda", dtype=torch.float32 if f32o else torch.bfloat16)
        bias = torch.randn(cout, device="cuda")
        ep = L.Epilogue(None, bias.data_ptr(), None, 0, int(f32o))
        run = lambda: L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))  # noqa: E731
        n = 1500 if B == 256 else 600
        for _ in range(n // 3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts[B] = e0.elapsed_time(e1) / n * 1000
    per256 = (ts[1024] - ts[256]) / 3
    print(f"{name:20s} B=256 {ts[256]:6.1f} us  B=1024 {ts[1024]:6.1f} us  fixed {ts[256] - per256:5.1f} us ({100 * (ts[256] - per256) / ts[1024]:.0f} % of the 1024-image launch)", flush=True)
