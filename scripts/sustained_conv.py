"""Sustained (hundreds of milliseconds) timing of the two 3x3 64->64 conv kernels: the card's power management reacts
over milliseconds, so a burst of ten launches (scripts/wreg_check.py) shows the kernel at a clock it cannot hold."""
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
B, hin, c = 1024, 32, 64
g = torch.Generator(device="cuda").manual_seed(0)
d = L.ConvDesc(B, hin, hin, c, c, 3, 3, 1, 1, 0)
w = (torch.randn(c, c, 3, 3, device="cuda", generator=g) * 0.05).bfloat16().float()
pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
L.check(lib.dvae_conv_pack(w.data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
x = torch.randn(B, hin, hin, c, device="cuda", generator=g).bfloat16()
if len(sys.argv) > 1 and sys.argv[1] == "relu":
    x = torch.relu(x)                                   # half zeros, like a post-ReLU activation
y = torch.empty_like(x)
scale = torch.rand(c, device="cuda", generator=g) + 0.5
shift = torch.randn(c, device="cuda", generator=g) * 0.2
ep = L.Epilogue(scale.data_ptr(), shift.data_ptr(), None, 1, 0)
fn = lambda: L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))  # noqa: E731
for rep in range(2):
    for wreg in (1, 0):
        L.check(lib.dvae_set_option(b"conv_wreg", wreg))
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
        ev[0].record()
        for k in range(30):
            for _ in range(100):
                fn()
            ev[k + 1].record()
        torch.cuda.synchronize()
        t = [ev[k].elapsed_time(ev[k + 1]) * 10 for k in range(30)]      # us per launch, per block of 100
        print(f"conv_wreg={wreg}: first 100 launches {t[0]:.1f} us, launches 400-500 {t[4]:.1f}, 1400-1500 {t[14]:.1f}, "
              f"last 100 {t[29]:.1f}; mean of the last 2000: {sum(t[10:]) / 20:.1f}", flush=True)
L.check(lib.dvae_set_option(b"conv_wreg", 1))
