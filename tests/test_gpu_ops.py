"""Per-kernel parity on the GPU: every C-ABI op against the same op of stock fp32 torch on the CPU
(the arithmetic the reference dispatches to).  fp32 path tolerance: 2e-5 of the tensor's scale for
forward results, 1e-4 for gradients that contract over thousands of pixels."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import dvae_amd
    from dvae_amd import _lib as L
    from dvae_amd.engine import ConvSpec
    from util_gpu import from_act, rel_err, to_act

CONVS = [
    # cin cout k stride pad transposed hin bias
    (3, 64, 3, 1, 1, False, 32, False),
    (64, 64, 3, 1, 1, False, 32, False),
    (64, 128, 3, 2, 1, False, 32, False),
    (64, 128, 1, 2, 0, False, 32, False),
    (128, 128, 3, 1, 1, False, 16, False),
    (128, 512, 1, 1, 0, False, 16, True),
    (256, 128, 1, 1, 0, False, 16, True),
    (128, 64, 4, 2, 1, True, 16, True),
    (64, 64, 3, 1, 1, True, 32, True),
    (64, 32, 3, 1, 1, True, 32, True),
    (32, 3, 3, 1, 1, True, 32, True),
    (64, 64, 3, 1, 1, False, 12, False),      # spatial size that is not a power of two
    (64, 128, 3, 2, 1, False, 13, False),     # odd size, strided: ragged dgrad phases
]


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("B", [2, 3])
def test_conv_fwd_dgrad_wgrad(cfg, B):
    cin, cout, k, stride, pad, tr, hin, bias = cfg
    lib = L.lib()
    g = torch.Generator().manual_seed(hash(cfg) & 0xffff)
    sp = ConvSpec("t", cin, cout, k, stride, pad, tr, hin, bias)
    w = torch.randn(sp.wshape, generator=g) / (cin * k * k) ** 0.5
    bvec = torch.randn(cout, generator=g) if bias else None
    x = torch.randn(B, cin, hin, hin, generator=g, requires_grad=True)
    w.requires_grad_(True)
    if tr:
        y = F.conv_transpose2d(x, w, bvec, stride=stride, padding=pad)
    else:
        y = F.conv2d(x, w, bvec, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    ho = y.shape[2]
    assert ho == sp.hout

    d = L.ConvDesc(B, hin, hin, cin, cout, k, k, stride, pad, int(tr))
    st = L.stream()
    wdev = w.detach().cuda().contiguous()
    pk = {}
    for op in (L.OP_FWD, L.OP_DGRAD):
        n = lib.dvae_conv_packed_elems(C.byref(d), op)
        pk[op] = torch.zeros(n, device="cuda")
        L.check(lib.dvae_conv_pack(wdev.data_ptr(), pk[op].data_ptr(), C.byref(d), op, L.F32, st))
    xa = to_act(x.detach())
    ya = torch.full((B, ho, ho, L.act_channels(cout)), 7.0, device="cuda")
    bdev = bvec.cuda() if bias else None
    ep = L.Epilogue(None, L.ptr(bdev), None, 0, 0)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk[L.OP_FWD].data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.F32, st))
    assert rel_err(from_act(ya, cout), y.detach()) < 2e-5
    if L.act_channels(cout) > cout:
        assert float(ya[..., cout:].abs().max()) == 0.0

    gya = to_act(gy)
    dxa = torch.full((B, hin, hin, L.act_channels(cin)), 3.0, device="cuda")
    L.check(lib.dvae_conv_dgrad(gya.data_ptr(), pk[L.OP_DGRAD].data_ptr(), dxa.data_ptr(), 0, C.byref(d), L.F32, st))
    assert rel_err(from_act(dxa, cin), x.grad) < 2e-5
    L.check(lib.dvae_conv_dgrad(gya.data_ptr(), pk[L.OP_DGRAD].data_ptr(), dxa.data_ptr(), 1, C.byref(d), L.F32, st))
    assert rel_err(from_act(dxa, cin), 2 * x.grad) < 2e-5

    ws = torch.empty(lib.dvae_conv_wgrad_workspace_bytes(C.byref(d)) // 4 + 4, device="cuda")
    dw = torch.full(sp.wshape, 9.0, device="cuda")
    db = torch.full((cout,), 9.0, device="cuda") if bias else None
    L.check(lib.dvae_conv_wgrad(xa.data_ptr(), gya.data_ptr(), dw.data_ptr(), L.ptr(db), ws.data_ptr(), ws.numel() * 4,
                                C.byref(d), L.F32, st))
    assert rel_err(dw.cpu(), w.grad) < 1e-4
    if bias:
        assert rel_err(db.cpu(), gy.sum(dim=(0, 2, 3))) < 1e-4


def test_conv_epilogue_scale_shift_residual_relu():
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    B, cin, cout, hin = 2, 64, 64, 16
    w = torch.randn(cout, cin, 3, 3, generator=g) / 24
    x = torch.randn(B, cin, hin, hin, generator=g)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(B, cout, hin, hin, generator=g)
    ref = F.relu(F.conv2d(x, w, None, 1, 1) * sc[None, :, None, None] + sh[None, :, None, None] + res)
    d = L.ConvDesc(B, hin, hin, cin, cout, 3, 3, 1, 1, 0)
    st = L.stream()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), 0), device="cuda")
    L.check(lib.dvae_conv_pack(w.cuda().data_ptr(), pk.data_ptr(), C.byref(d), 0, L.F32, st))
    ya = torch.empty(B, hin, hin, cout, device="cuda")
    scd, shd, resa = sc.cuda(), sh.cuda(), to_act(res)
    ep = L.Epilogue(scd.data_ptr(), shd.data_ptr(), resa.data_ptr(), 1, 0)
    L.check(lib.dvae_conv_fwd(to_act(x).data_ptr(), pk.data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.F32, st))
    assert rel_err(from_act(ya, cout), ref) < 2e-5


@pytest.mark.parametrize("groups,rows,c", [(1, 2 * 32 * 32, 64), (1, 3 * 16 * 16, 128), (3, 32 * 32, 64), (2, 32 * 32, 32),
                                            (1, 5000, 64)])
@pytest.mark.parametrize("relu", [0, 1])
def test_norm_fwd_bwd(groups, rows, c, relu):
    """BatchNorm (groups == 1, affine, running stats) and InstanceNorm (groups == B) against torch."""
    lib = L.lib()
    g = torch.Generator().manual_seed(groups * 1000 + c + relu)
    x = (torch.randn(groups * rows, c, generator=g) * 2 + 3).requires_grad_(True)
    res = torch.randn(groups * rows, c, generator=g).requires_grad_(True)
    bn = groups == 1
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(c, generator=g).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    xg = x.view(groups, rows, c)
    mean = xg.mean(1, keepdim=True)
    var = xg.var(1, unbiased=False, keepdim=True)
    xhat = (xg - mean) / torch.sqrt(var + 1e-5)
    y = (xhat * gamma + beta if bn else xhat).reshape(groups * rows, c)
    if bn:
        y = y + res
    if relu:
        y = F.relu(y)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)

    st = L.stream()
    ws = torch.empty(lib.dvae_norm_workspace_bytes(groups, rows, c) // 4 + 4, device="cuda")
    xd, resd, yd = x.detach().cuda(), res.detach().cuda(), torch.empty(groups * rows, c, device="cuda")
    sm, si = torch.empty(groups, c, device="cuda"), torch.empty(groups, c, device="cuda")
    gd, bd, rmd, rvd = gamma.detach().cuda(), beta.detach().cuda(), rm.cuda(), rv.cuda()
    L.check(lib.dvae_norm_fwd_train(xd.data_ptr(), resd.data_ptr() if bn else None, yd.data_ptr(),
                                    gd.data_ptr() if bn else None, bd.data_ptr() if bn else None,
                                    rmd.data_ptr() if bn else None, rvd.data_ptr() if bn else None,
                                    sm.data_ptr(), si.data_ptr(), ws.data_ptr(), ws.numel() * 4, groups, rows, c, relu,
                                    0.1, 1e-5, L.F32, st))
    assert rel_err(yd.cpu(), y.detach()) < 2e-5
    assert rel_err(sm.cpu(), mean.reshape(groups, c)) < 2e-5
    if bn:
        assert rel_err(rmd.cpu(), 0.9 * rm + 0.1 * mean.reshape(c)) < 2e-5
        assert rel_err(rvd.cpu(), 0.9 * rv + 0.1 * xg.var(1, unbiased=True).reshape(c)) < 2e-5
    dx, dres = torch.empty_like(xd), torch.empty_like(xd)
    dg, db = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    gyd = gy.cuda()
    L.check(lib.dvae_norm_bwd(gyd.data_ptr(), yd.data_ptr(), xd.data_ptr(), dx.data_ptr(), dres.data_ptr() if bn else None,
                              gd.data_ptr() if bn else None, sm.data_ptr(), si.data_ptr(),
                              dg.data_ptr() if bn else None, db.data_ptr() if bn else None, ws.data_ptr(), ws.numel() * 4,
                              groups, rows, c, relu, L.F32, st))
    assert rel_err(dx.cpu(), x.grad) < 1e-4
    if bn:
        assert rel_err(dres.cpu(), res.grad) < 1e-6
        assert rel_err(dg.cpu(), gamma.grad) < 1e-4
        assert rel_err(db.cpu(), beta.grad) < 1e-4


def test_bn_eval_matches_torch():
    lib = L.lib()
    g = torch.Generator().manual_seed(3)
    rows, c = 2048, 128
    x, res = torch.randn(rows, c, generator=g), torch.randn(rows, c, generator=g)
    ga, be = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    ref = F.relu(F.batch_norm(x, rm, rv, ga, be, False, 0.1, 1e-5) + res)
    y = torch.empty(rows, c, device="cuda")
    ss = torch.empty(2 * c, device="cuda")
    t = [v.cuda() for v in (x, res, ga, be, rm, rv)]
    L.check(lib.dvae_bn_fwd_eval(t[0].data_ptr(), t[1].data_ptr(), y.data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                 t[4].data_ptr(), t[5].data_ptr(), ss.data_ptr(), rows, c, 1, 1e-5, L.F32, L.stream()))
    assert rel_err(y.cpu(), ref) < 2e-5


@pytest.mark.parametrize("sid", [1, 0x80000001], ids=["24-bit uniforms", "16-bit uniforms"])
def test_philox_normal_moments_and_index_keying(sid):
    """Stream ids with bit 31 set draw eight normals per Philox call from 16-bit uniforms (the bf16 inference path's
    noise): same keying, same moments, |eps| <= sqrt(-2 ln 2^-16) = 4.71."""
    lib = L.lib()
    B, per = 8, 65536
    a = torch.empty(B, per, device="cuda")
    L.check(lib.dvae_philox_normal(a.data_ptr(), None, 100, B, per, 1234, sid, L.stream()))
    if sid >> 31:
        assert float(a.abs().max()) <= 4.7096 + 1e-4
        other = torch.empty(B, per, device="cuda")
        L.check(lib.dvae_philox_normal(other.data_ptr(), None, 100, B, per, 1234, sid & 0x7fffffff, L.stream()))
        assert not torch.equal(other, a)
    v = a.cpu().double()
    assert abs(float(v.mean())) < 5e-3 and abs(float(v.var()) - 1) < 1e-2
    assert abs(float((v ** 4).mean()) - 3) < 0.1
    # image 103 drawn alone equals row 3 of the batch: the stream is keyed by the global image index
    b = torch.empty(1, per, device="cuda")
    L.check(lib.dvae_philox_normal(b.data_ptr(), None, 103, 1, per, 1234, sid, L.stream()))
    assert torch.equal(b[0], a[3])
    idx = torch.tensor([107, 100], dtype=torch.int32, device="cuda")
    c2 = torch.empty(2, per, device="cuda")
    L.check(lib.dvae_philox_normal(c2.data_ptr(), idx.data_ptr(), 0, 2, per, 1234, sid, L.stream()))
    assert torch.equal(c2[0], a[7]) and torch.equal(c2[1], a[0])
    L.check(lib.dvae_philox_normal(b.data_ptr(), None, 103, 1, per, 1234, sid + 1, L.stream()))
    assert not torch.equal(b[0], a[3])


def test_quantiser_known_answers(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "g5_quant_kat.npz"))
    lib = L.lib()
    x = torch.from_numpy(g["x"])
    n = x.numel()
    hw = n
    a = torch.zeros(1, hw, 4, device="cuda")
    a[0, :, 0] = x.cuda()
    a[0, :, 1] = x.cuda()
    a[0, :, 2] = x.cuda()
    out = torch.zeros(1, hw, 3, dtype=torch.uint8, device="cuda")
    L.check(lib.dvae_purify_quant(a.data_ptr(), None, 0, None, out.data_ptr(), 0, 1, hw, L.F32, L.stream()))
    assert np.array_equal(out[0, :, 1].cpu().numpy(), g["q"])


@pytest.mark.parametrize("scale", [1.0, 100.0])
def test_clip_adam_matches_torch(scale):
    """clip_grad_norm_(5.0) + torch.optim.Adam(lr 5e-3) for 3 steps on a flat bucket; scale=100 forces
    the clip branch.  Element-wise parity (gradients here are O(1), no near-zero sign ambiguity)."""
    lib = L.lib()
    g = torch.Generator().manual_seed(11)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=5e-3, weight_decay=0.0)
    pd, md, vd = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    hyper = torch.tensor([5e-3, 0.9, 0.999, 1e-8, 5.0, 1.0, 0.0, 0.0], device="cuda")
    partial = torch.zeros(L.NPARTIAL, device="cuda")
    scal = torch.zeros(L.SCAL_COUNT, device="cuda")
    for step in range(3):
        gr = (torch.rand(n, generator=g) + 0.5) * torch.sign(torch.randn(n, generator=g)) * scale / n ** 0.5
        ref.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 5.0)
        opt.step()
        gd = gr.cuda()
        L.check(lib.dvae_sqnorm_partial(gd.data_ptr(), n, partial.data_ptr(), L.stream()))
        L.check(lib.dvae_clip_adam(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, hyper.data_ptr(),
                                   partial.data_ptr(), scal.data_ptr(), L.stream()))
        assert abs(float(scal[L.SCAL["grad_norm"]]) - float(tn)) <= 1e-5 * float(tn)
        assert rel_err(pd.cpu(), ref.detach()) < 1e-6
    assert float(hyper[6]) == 3.0


def test_reparam_kl_and_recon_loss_kernels():
    """K3 / K4 against their torch formulas (main_vae_disentangle.py:190-192, :204; resnet.py:2257-2259)."""
    lib = L.lib()
    g = torch.Generator().manual_seed(21)
    rows, Ld = 3 * 16 * 16, 256
    mulv = torch.randn(rows, 2 * Ld, generator=g) * 0.7
    eps = torch.randn(rows, Ld, generator=g)
    mu, lv = mulv[:, :Ld].clone().requires_grad_(True), mulv[:, Ld:].clone().requires_grad_(True)
    z1 = eps * torch.exp(0.5 * lv) + mu
    kl_sum = (-0.5 * torch.sum(1 + lv - mu ** 2 - lv.exp(), dim=1)).sum()
    dz = torch.randn(rows, Ld, generator=g)
    ckl = 0.01 / rows
    (z1 * dz).sum().backward(retain_graph=True)
    (kl_sum * ckl).backward()
    md, ed = mulv.cuda(), eps.cuda()
    z1d = torch.empty(rows, Ld, device="cuda")
    part = torch.zeros(L.NPARTIAL, device="cuda")
    L.check(lib.dvae_reparam_kl_fwd(md.data_ptr(), ed.data_ptr(), z1d.data_ptr(), part.data_ptr(), rows, Ld, L.F32, L.stream()))
    assert rel_err(z1d.cpu(), z1.detach()) < 1e-6
    assert abs(float(part.double().sum()) - float(kl_sum)) <= 1e-5 * abs(float(kl_sum))
    scal = torch.zeros(L.SCAL_COUNT, device="cuda")
    scal[L.SCAL["c_kl"]] = ckl
    dml = torch.empty(rows, 2 * Ld, device="cuda")
    L.check(lib.dvae_reparam_kl_bwd(dz.cuda().data_ptr(), md.data_ptr(), ed.data_ptr(), scal.data_ptr(), dml.data_ptr(), rows, Ld,
                                    L.F32, L.stream()))
    assert rel_err(dml[:, :Ld].cpu(), mu.grad) < 2e-6
    assert rel_err(dml[:, Ld:].cpu(), lv.grad) < 2e-6

    npix = 2 * 32 * 32
    rp = (torch.randn(npix, 3, generator=g) * 2).requires_grad_(True)
    p = (torch.randn(npix, 3, generator=g) * 0.5).requires_grad_(True)
    x = torch.rand(npix, 3, generator=g)
    rec = torch.sigmoid(rp)
    l0 = torch.nn.functional.mse_loss(rec, x)
    l1 = torch.nn.functional.mse_loss((rec + p).clamp(0, 1), x)
    l2 = torch.nn.functional.mse_loss(p, torch.zeros_like(p))
    (l0 + l1 + l2).backward()

    def pad(t):
        o = torch.zeros(npix, 4)
        o[:, :3] = t.detach()
        return o.cuda()
    rpd, pdv, xd = pad(rp), pad(p), pad(x)
    recd = torch.empty(npix, 4, device="cuda")
    part3 = torch.zeros(3 * L.NPARTIAL, device="cuda")
    L.check(lib.dvae_recon_loss_fwd(rpd.data_ptr(), pdv.data_ptr(), xd.data_ptr(), recd.data_ptr(), part3.data_ptr(), npix, L.F32, L.stream()))
    sums = torch.zeros(4, device="cuda")
    L.check(lib.dvae_loss_reduce(part3.data_ptr(), part.data_ptr(), sums.data_ptr(), L.stream()))
    L.check(lib.dvae_loss_gate(sums.data_ptr(), scal.data_ptr(), float(npix * 3), float(rows), float(npix * 3), float(rows),
                               1.0, 0.01, 1, L.stream()))
    sc = scal.cpu()
    assert abs(float(sc[L.SCAL["loss_rec"]]) - float(l0)) < 1e-6 and abs(float(sc[L.SCAL["loss_rec_p"]]) - float(l1)) < 1e-6
    assert abs(float(sc[L.SCAL["loss_p"]]) - float(l2)) < 1e-6
    assert rel_err(recd[:, :3].cpu(), rec.detach()) < 1e-6
    drp, dp = torch.empty(npix, 4, device="cuda"), torch.empty(npix, 4, device="cuda")
    L.check(lib.dvae_recon_loss_bwd(recd.data_ptr(), pdv.data_ptr(), xd.data_ptr(), scal.data_ptr(), drp.data_ptr(), dp.data_ptr(),
                                    npix, L.F32, L.stream()))
    assert rel_err(drp[:, :3].cpu(), rp.grad) < 2e-6
    assert rel_err(dp[:, :3].cpu(), p.grad) < 2e-6
