"""Time one bf16 conv entry point at a given shape on the current library build.

usage: python scripts/conv_time.py cin cout k stride pad transposed hin batch [dgrad|wgrad]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402


def main():
    cin, cout, k, stride, pad, tr, hin, B = (int(v) for v in sys.argv[1:9])
    dgrad = len(sys.argv) > 9 and sys.argv[9] == "dgrad"
    wgrad = len(sys.argv) > 9 and sys.argv[9] == "wgrad"
    lib = L.lib()
    d = L.ConvDesc(B, hin, hin, cin, cout, k, k, stride, pad, tr)
    ho = (hin - 1) * stride - 2 * pad + k if tr else (hin + 2 * pad - k) // stride + 1
    st = L.stream()
    op = L.OP_DGRAD if dgrad else L.OP_FWD
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device="cuda") * 0.05
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), op), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.data_ptr(), pk.data_ptr(), C.byref(d), op, L.BF16, st))
    x = torch.randn(B, hin, hin, L.act_channels(cin, L.BF16), device="cuda").bfloat16()
    y = torch.empty(B, ho, ho, L.act_channels(cout, L.BF16), device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(cout, device="cuda")
    ep = L.Epilogue(None, bias.data_ptr(), None, 0, 0)

    dw = torch.empty_like(w)
    db = torch.empty(cout, device="cuda")
    ws = torch.empty(lib.dvae_conv_wgrad_workspace_bytes(C.byref(d)) // 4 + 16, device="cuda")

    def run():
        if wgrad:
            L.check(lib.dvae_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                        C.byref(d), L.BF16, st))
        elif dgrad:
            L.check(lib.dvae_conv_dgrad(y.data_ptr(), pk.data_ptr(), x.data_ptr(), 0, C.byref(d), L.BF16, st))
        else:
            L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{' '.join(sys.argv[1:])}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")


if __name__ == "__main__":
    main()
