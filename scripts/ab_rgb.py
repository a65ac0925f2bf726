"""A/B of the 32 -> 3 channel decoder head (fp32 output, fused input normalisation): the dedicated kernel against the
halo kernel, switched inside one process with dvae_set_option, sustained back-to-back launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402

lib = L.lib()
st = L.stream()
for B in (128, 1024):
    d = L.ConvDesc(B, 32, 32, 32, 3, 3, 3, 1, 1, 1)
    w = torch.randn(32, 3, 3, 3, device="cuda") * 0.05
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    x = torch.randn(B, 32, 32, 32, device="cuda").bfloat16()
    ss = torch.randn(B, 2, 32, device="cuda")
    y = [torch.empty(B, 32, 32, 4, device="cuda") for _ in range(2)]
    bias = torch.randn(3, device="cuda")
    for rep in range(2):
        for opt in (0, 1):
            L.check(lib.dvae_set_option(b"conv_rgb", opt))
            ep = L.Epilogue(None, bias.data_ptr(), None, 0, 1, ss.data_ptr(), 1)
            run = lambda: L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), y[opt].data_ptr(), C.byref(d), L.BF16, st))  # noqa: E731
            n = 1000
            for _ in range(300):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f"B={B} conv_rgb={opt}: {e0.elapsed_time(e1) / n * 1000:6.1f} us", flush=True)
    print("max |new - old| =", float((y[0] - y[1]).abs().max()), " scale", float(y[0].abs().max()), flush=True)
