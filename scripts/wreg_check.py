"""Correctness and A/B timing of the weights-in-registers 3x3 conv kernel against the weights-in-LDS kernel and torch."""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd  # noqa: E402,F401
from dvae_amd import _lib as L  # noqa: E402

lib = L.lib()
st = L.stream()


def run(cin, cout, hin, B, tr, mode, seed=0, time_it=False):
    """mode: 'plain' | 'bn_relu' | 'res_relu' | 'dgrad' | 'dgrad_acc'"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    d = L.ConvDesc(B, hin, hin, cin, cout, 3, 3, 1, 1, tr)
    dgrad = mode.startswith("dgrad")
    op = L.OP_DGRAD if dgrad else L.OP_FWD
    w = (torch.randn((cin, cout, 3, 3) if tr else (cout, cin, 3, 3), device="cuda", generator=g) * 0.05).bfloat16().float()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), op), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.data_ptr(), pk.data_ptr(), C.byref(d), op, L.BF16, st))
    ci, co = (cout, cin) if dgrad else (cin, cout)         # channels of the tensor read / written
    cis, cos = L.act_channels(ci, L.BF16), L.act_channels(co, L.BF16)
    x = torch.zeros(B, hin, hin, cis, device="cuda", dtype=torch.bfloat16)
    x[..., :ci] = torch.randn(B, hin, hin, ci, device="cuda", generator=g).bfloat16()
    res = torch.zeros(B, hin, hin, cos, device="cuda", dtype=torch.bfloat16)
    res[..., :co] = torch.randn(B, hin, hin, co, device="cuda", generator=g).bfloat16()
    scale = torch.rand(co, device="cuda", generator=g) + 0.5
    shift = torch.randn(co, device="cuda", generator=g) * 0.2
    outs = {}
    for wreg in (1, 0):
        L.check(lib.dvae_set_option(b"conv_wreg", wreg))
        y = res.clone() if mode == "dgrad_acc" else torch.full((B, hin, hin, cos), 7.0, device="cuda", dtype=torch.bfloat16)
        if dgrad:
            fn = lambda: L.check(lib.dvae_conv_dgrad(x.data_ptr(), pk.data_ptr(), y.data_ptr(), int(mode == "dgrad_acc"), C.byref(d), L.BF16, st))  # noqa: E731
        else:
            ep = L.Epilogue(scale.data_ptr() if mode != "plain" else None, shift.data_ptr(),
                            res.data_ptr() if mode == "res_relu" else None, int(mode != "plain"), 0)
            fn = lambda: L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))  # noqa: E731
        fn()
        torch.cuda.synchronize()
        outs[wreg] = y.clone()
        if time_it and mode != "dgrad_acc":
            ts = []
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1000)
            outs[("t", wreg)] = min(ts)
    L.check(lib.dvae_set_option(b"conv_wreg", 1))
    # torch reference on the same bf16-rounded operands, fp32 math
    xf = x[..., :ci].float().permute(0, 3, 1, 2)
    if dgrad:
        ref = F.conv_transpose2d(xf, w, None, 1, 1) if not tr else F.conv2d(xf, w, None, 1, 1)
    else:
        ref = F.conv_transpose2d(xf, w, None, 1, 1) if tr else F.conv2d(xf, w, None, 1, 1)
    rf = res[..., :co].float().permute(0, 3, 1, 2)
    if mode == "plain":
        ref = ref + shift[None, :, None, None]
    elif mode == "bn_relu":
        ref = F.relu(ref * scale[None, :, None, None] + shift[None, :, None, None])
    elif mode == "res_relu":
        ref = F.relu(ref * scale[None, :, None, None] + shift[None, :, None, None] + rf)
    elif mode == "dgrad_acc":
        ref = ref + rf
    ref = ref.permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    e1 = float((outs[1][..., :co].float() - ref).abs().max()) / sc
    e0 = float((outs[0][..., :co].float() - ref).abs().max()) / sc
    pad = float(outs[1][..., co:].float().abs().max()) if cos > co else 0.0
    msg = f"cin {cin} cout {cout} hin {hin} B {B} tr {tr} {mode}: wreg err {e1:.2e} halo err {e0:.2e} pad {pad}"
    if time_it and ("t", 1) in outs:
        msg += f"  time wreg {outs[('t', 1)]:.1f} us  halo {outs[('t', 0)]:.1f} us"
    print(msg, flush=True)
    return e1 < 8e-3 and pad == 0.0


ok = True
for cin, cout, hin in ((64, 64, 32), (64, 32, 32), (32, 64, 32), (32, 32, 32), (64, 64, 64), (32, 3, 32)):
    for B in (3, 300):
        for tr in (0, 1):
            for mode in ("plain", "bn_relu", "res_relu", "dgrad", "dgrad_acc"):
                if cout == 3 and not mode.startswith("dgrad"):
                    continue                                # 3-channel outputs are fp32 heads: not this kernel
                ok &= run(cin, cout, hin, B, tr, mode)
print("ALL OK" if ok else "FAILED", flush=True)
if ok and len(sys.argv) > 1:
    for cin, cout in ((64, 64), (64, 32), (32, 32)):
        for mode in ("bn_relu", "res_relu", "dgrad"):
            run(cin, cout, 32, 1024, 0, mode, time_it=True)
    run(64, 64, 32, 128, 0, "bn_relu", time_it=True)
    run(64, 64, 64, 256, 0, "bn_relu", time_it=True)
