"""A/B of the 3x3 conv 128 -> 128 (layer2's second conv, with scale / shift / residual / ReLU): the weights-in-registers kernel against the
image-resident kernel, switched inside one process with dvae_set_option, sustained back-to-back launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402

lib = L.lib()
st = L.stream()
HIN = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for B in ((128, 1024) if HIN == 16 else (max(1, 32768 // (HIN * HIN)), max(1, 262144 // (HIN * HIN)))):
    d = L.ConvDesc(B, HIN, HIN, 128, 128, 3, 3, 1, 1, 0)
    w = torch.randn(128, 128, 3, 3, device="cuda") * 0.05
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    x = torch.relu(torch.randn(B, HIN, HIN, 128, device="cuda")).bfloat16()
    y = [torch.empty(B, HIN, HIN, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    sc, sh = torch.rand(128, device="cuda") + 0.5, torch.randn(128, device="cuda")
    res = torch.randn(B, HIN, HIN, 128, device="cuda").bfloat16()
    for rep in range(2):
        for opt in (0, 1):
            L.check(lib.dvae_set_option(b"conv_mid", opt))
            ep = L.Epilogue(sc.data_ptr(), sh.data_ptr(), res.data_ptr(), 1, 0)
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
            t = e0.elapsed_time(e1) / n * 1000
            print(f"B={B} conv_mid={opt}: {t:6.1f} us  {75497472 * B * (HIN * HIN // 256) / t / 1e6:7.1f} TFLOP/s", flush=True)
    print("max |new - old| =", float((y[0].float() - y[1].float()).abs().max()), " scale", float(y[0].float().abs().max()), flush=True)
