"""dvae_norm_fwd_train_static (the entry point that allocates nothing: its partial statistics live in a scratch array inside
the library's device image) against dvae_norm_fwd_train with a caller-provided workspace, through the C ABI.  Same
kernels, so every output must be bit-identical -- on two calls in a row, the second with a smaller input (the scratch
still holds the first call's contents), and for both dtypes.  Sizes beyond the scratch are rejected without a launch."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import dvae_amd
    from dvae_amd import _lib as L


def _run(lib, static, x, res, gamma, beta, groups, rows, c, relu, dtype):
    """One forward through either entry point; returns every output (y, running stats, saved stats)."""
    f32 = dict(device="cuda", dtype=torch.float32)
    y = torch.full_like(x, 7.0)
    bn = gamma is not None
    rm = torch.linspace(-0.1, 0.1, c, **f32) if bn else None
    rv = torch.linspace(0.5, 1.5, c, **f32) if bn else None
    mean = torch.zeros(groups, c, **f32)
    invstd = torch.zeros(groups, c, **f32)
    st = L.stream()
    if static:
        rc = lib.dvae_norm_fwd_train_static(L.ptr(x), L.ptr(res), L.ptr(y), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv),
                                            L.ptr(mean), L.ptr(invstd), groups, rows, c, relu, 0.1, 1e-5, dtype, st)
    else:
        nb = lib.dvae_norm_workspace_bytes(groups, rows, c)
        ws = torch.empty(nb // 4 + 16, **f32)
        rc = lib.dvae_norm_fwd_train(L.ptr(x), L.ptr(res), L.ptr(y), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv),
                                     L.ptr(mean), L.ptr(invstd), L.ptr(ws), ws.numel() * 4, groups, rows, c, relu, 0.1, 1e-5,
                                     dtype, st)
    L.check(rc)
    torch.cuda.synchronize()
    return [t for t in (y, rm, rv, mean, invstd) if t is not None]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_static_scratch_norm_equals_workspace_norm_on_two_calls_in_a_row(dtype):
    lib = L.lib()
    dt = L.F32 if dtype == "f32" else L.BF16
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    # call 1: BatchNorm over 8 images of 32x32x64 with residual + ReLU; call 2 (smaller): 2 images of 16x16x128, no
    # residual, no ReLU; call 3: InstanceNorm + ReLU of 3 images (fp32: three-kernel path through the scratch)
    cases = [(1, 8 * 32 * 32, 64, True, True, 1), (1, 2 * 16 * 16, 128, True, False, 0), (3, 32 * 32, 32, False, False, 1)]
    for groups, rows, c, bn, with_res, relu in cases:
        x = (torch.randn(groups * rows, c, generator=g, device="cuda") * 1.5 + 0.3).to(tdt)
        res = torch.randn(groups * rows, c, generator=g, device="cuda").to(tdt) if with_res else None
        gamma = torch.rand(c, generator=g, device="cuda") + 0.5 if bn else None
        beta = torch.rand(c, generator=g, device="cuda") - 0.5 if bn else None
        want = _run(lib, False, x, res, gamma, beta, groups, rows, c, relu, dt)
        got = _run(lib, True, x, res, gamma, beta, groups, rows, c, relu, dt)
        assert len(want) == len(got)
        for a, b in zip(want, got):
            assert torch.equal(a, b), (groups, rows, c)
        # and the result is a normalisation: per-channel mean ~ beta (or 0), checked on the no-ReLU BatchNorm case
        if bn and not relu:
            y = got[0].float()
            assert float((y.mean(0) - beta).abs().max()) < 2e-2


def test_static_scratch_norm_rejects_what_does_not_fit():
    lib = L.lib()
    rows, c = 512 * 256, 256                       # 512 chunks x 3 x 256 channels = 393 216 floats > 262 144
    assert lib.dvae_norm_workspace_bytes(1, rows, c) > 262144 * 4
    x = torch.zeros(8, c, device="cuda")           # never read: the call must fail before any launch
    y = torch.full_like(x, 3.0)
    s = torch.zeros(c, device="cuda")
    rc = lib.dvae_norm_fwd_train_static(L.ptr(x), None, L.ptr(y), None, None, None, None, L.ptr(s), L.ptr(s), 1, rows, c, 0,
                                        0.1, 1e-5, L.F32, L.stream())
    assert rc == -4 and b"scratch" in lib.dvae_last_error()
    torch.cuda.synchronize()
    assert float(y.min()) == 3.0
    # InstanceNorm of 300 images of 32x32x64 needs 300 x 17 x 64 floats: too many as well
    rc = lib.dvae_norm_fwd_train_static(L.ptr(x), None, L.ptr(y), None, None, None, None, L.ptr(s), L.ptr(s), 300, 1024, 64, 1,
                                        0.0, 1e-5, L.F32, L.stream())
    assert rc == -4
