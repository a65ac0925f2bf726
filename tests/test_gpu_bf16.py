"""bf16 MFMA path (bf16 activations / packed weights, fp32 accumulation, fp32 statistics, losses,
gradients of parameters and optimiser state) against the fp32 references.

Stated tolerance: bf16 keeps 8 significant bits (2^-9 = 2e-3 relative rounding per stored value), so
conv outputs are held to 6e-3 of the tensor's scale (1e-4 with fp32 output), whole-network outputs to 6e-2
in L2, loss scalars to 3e-2, the gradient bucket to cosine > 0.998 with the fp32 oracle (single tensors up to
0.25 in L2 at batch 8: at initialisation the KL term dominates the stem gradient and BatchNorm's backward
cancels most of it, which amplifies the bf16 rounding of the stored gradient maps), and purified uint8
images to a mean difference < 0.75 LSB (> 98.5 % of pixels within 4 LSB) of the fp32 path."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dvae_oracle as O

if torch.cuda.is_available():
    import dvae_amd
    from dvae_amd import _lib as L
    from dvae_amd.engine import ConvSpec, DVAEConfig, DVAEEngine
    from util_gpu import from_act, l2_err, rel_err, to_act

CONVS = [
    (3, 64, 3, 1, 1, False, 32, False),
    (64, 64, 3, 1, 1, False, 32, False),
    (64, 128, 3, 2, 1, False, 32, False),
    (64, 128, 1, 2, 0, False, 32, False),
    (128, 512, 1, 1, 0, False, 16, True),
    (128, 64, 4, 2, 1, True, 16, True),
    (64, 32, 3, 1, 1, True, 32, True),
    (32, 3, 3, 1, 1, True, 32, True),
    (64, 128, 3, 2, 1, False, 13, False),
    (64, 64, 3, 1, 1, False, 16, False),      # LDS-halo kernel with 16-wide tiles
    (32, 64, 3, 1, 1, False, 64, False),      # 64x64 images: two tiles per row
    (64, 64, 3, 1, 1, True, 32, True),
    (64, 64, 3, 1, 1, False, 64, False),      # 64x64 images: all-tap weight gradient with two tiles per row
    (256, 128, 1, 1, 0, False, 16, True),     # post_conv: weights-resident 1x1 kernel
    (128, 128, 3, 1, 1, False, 16, False),    # layer2's second conv: image-resident kernel, two blocks of 64 channels
    # tile grids that are not powers of two (224 = 7 x 32 columns, 28 x 8 rows; 96 = 3 x 32, 12 x 8): the persistent
    # kernels split a tile index by multiply-high instead of shifts
    (64, 64, 3, 1, 1, False, 96, False),      # weights-in-registers conv both ways, all-tap weight gradient
    (64, 32, 3, 1, 1, True, 96, True),
    (32, 3, 3, 1, 1, True, 96, True),         # LDS-weights kernel, 32 stored input channels
    (3, 64, 3, 1, 1, False, 96, False),       # first conv
    (64, 64, 3, 1, 1, False, 224, False),
    # the classifier's layer3 / layer4 shapes (resnet.py:62-122)
    (128, 256, 3, 2, 1, False, 16, False),
    (256, 256, 3, 1, 1, False, 8, False),
    (128, 256, 1, 2, 0, False, 16, False),
    (256, 512, 3, 2, 1, False, 8, False),
    (512, 512, 3, 1, 1, False, 4, False),
    (256, 512, 1, 2, 0, False, 8, False),
    # the tile-resident kernel on maps larger than one 16x16 tile (64x64 / 96x96 / 224x224 images): halos come from the
    # neighbouring tiles, zeros only outside the map; 3 x 3 and 7 x 7 tile grids split the tile index by multiply-high
    (128, 64, 4, 2, 1, True, 32, True),       # transposed conv, 16 slabs
    (128, 64, 4, 2, 1, True, 48, True),
    (128, 128, 3, 1, 1, False, 32, False),    # 18 slabs
    (128, 128, 3, 1, 1, False, 112, False),
    (64, 128, 3, 2, 1, False, 64, False),     # its input-gradient: 9 slabs over the 32x32 gradient map
    (64, 128, 3, 2, 1, False, 96, False),
]


@pytest.mark.parametrize("cfg", CONVS)
def test_conv_bf16(cfg):
    cin, cout, k, stride, pad, tr, hin, bias = cfg
    B = 3
    lib = L.lib()
    g = torch.Generator().manual_seed(hash(cfg) & 0xffff)
    sp = ConvSpec("t", cin, cout, k, stride, pad, tr, hin, bias)
    q = lambda t: t.bfloat16().float()      # noqa: E731  operands as the kernel sees them
    w = q(torch.randn(sp.wshape, generator=g) / (cin * k * k) ** 0.5).requires_grad_(True)
    bvec = torch.randn(cout, generator=g) if bias else None
    x = q(torch.randn(B, cin, hin, hin, generator=g)).requires_grad_(True)
    y = F.conv_transpose2d(x, w, bvec, stride=stride, padding=pad) if tr else F.conv2d(x, w, bvec, stride=stride, padding=pad)
    gy = q(torch.randn(y.shape, generator=g))
    y.backward(gy)
    ho = y.shape[2]
    d = L.ConvDesc(B, hin, hin, cin, cout, k, k, stride, pad, int(tr))
    st = L.stream()
    wdev = w.detach().cuda().contiguous()
    pk = {}
    for op in (L.OP_FWD, L.OP_DGRAD):
        pk[op] = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), op), device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_conv_pack(wdev.data_ptr(), pk[op].data_ptr(), C.byref(d), op, L.BF16, st))
    xa = to_act(x.detach(), L.BF16)
    ya = torch.full((B, ho, ho, L.act_channels(cout, L.BF16)), 7.0, device="cuda", dtype=torch.bfloat16)
    bdev = bvec.cuda() if bias else None
    ep = L.Epilogue(None, L.ptr(bdev), None, 0, 0)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk[L.OP_FWD].data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.BF16, st))
    assert rel_err(from_act(ya, cout), y.detach()) < 6e-3          # exact products, fp32 sums, one bf16 rounding of y
    yf = torch.empty(B, ho, ho, L.act_channels(cout, L.F32), device="cuda")     # fp32 outputs use the fp32 channel padding
    ep = L.Epilogue(None, L.ptr(bdev), None, 0, 1)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk[L.OP_FWD].data_ptr(), C.byref(ep), yf.data_ptr(), C.byref(d), L.BF16, st))
    assert rel_err(from_act(yf, cout), y.detach()) < 1e-4          # fp32 output: only summation order differs
    gya = to_act(gy, L.BF16)
    dxa = torch.zeros(B, hin, hin, L.act_channels(cin, L.BF16), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_dgrad(gya.data_ptr(), pk[L.OP_DGRAD].data_ptr(), dxa.data_ptr(), 0, C.byref(d), L.BF16, st))
    assert rel_err(from_act(dxa, cin), x.grad) < 6e-3
    ws = torch.empty(lib.dvae_conv_wgrad_workspace_bytes(C.byref(d)) // 4 + 4, device="cuda")
    dw = torch.full(sp.wshape, 9.0, device="cuda")
    db = torch.full((cout,), 9.0, device="cuda") if bias else None
    L.check(lib.dvae_conv_wgrad(xa.data_ptr(), gya.data_ptr(), dw.data_ptr(), L.ptr(db), ws.data_ptr(), ws.numel() * 4,
                                C.byref(d), L.BF16, st))
    assert rel_err(dw.cpu(), w.grad) < 1e-4                        # fp32 accumulation of exact bf16 products
    if bias:
        assert rel_err(db.cpu(), gy.sum(dim=(0, 2, 3))) < 1e-4


def _engine(dtype, B, ov=None):
    ocfg, cfg = O.DVAEConfig(), DVAEConfig()
    eng = DVAEEngine(cfg, B, "cuda", dtype)
    p = O.recipe_params(ocfg, ov)
    eng.load_params(p)
    return ocfg, eng, p


@pytest.mark.parametrize("train", [True, False])
def test_model_forward_bf16(train):
    ocfg, eng, p = _engine(L.BF16, 4)
    x, y, eps = O.make_inputs(ocfg, 4)
    with torch.no_grad():
        ref = O.forward(p, ocfg, x, y, eps, train, {})
    x0 = eng.input_from_nchw(x.cuda())
    e = eng.eps_from_nchw(eps.cuda())
    rec_pre, pp = eng.forward(x0, y.cuda(), e, 4, train)
    rec = eng.loss_sums()
    mulv = eng.act("mulv", 16, 512, f32=True)
    assert l2_err(from_act(rec[:4], 3), ref["rec"]) < 3e-2
    assert l2_err(from_act(pp[:4], 3), ref["perturbation"]) < 6e-2
    assert l2_err(from_act(mulv[:4, :, :, :256], 256), ref["mu"]) < 6e-2
    assert l2_err(from_act(mulv[:4, :, :, 256:], 256), ref["log_var"]) < 6e-2


@pytest.mark.parametrize("B", [8, 128])
def test_train_step_bf16_tracks_fp32(B):
    """Losses and gradients of one step of the bf16 engine (gate open, epoch 11, stage 1), at 8 images and at the
    reference's batch of 128 (configs/cifar10/resnet18DVAE.yaml), where the backward kernels loop over tiles.
    The bounds are derived from bf16 rounding itself: the oracle runs a second time with every tensor the device path
    stores in bf16 (activations, gradient maps, packed weights, the embedding table) rounded at the same points
    (oracle.bf16_storage) and fp32 everywhere else.  Against that emulation the loss scalars must agree to 2e-4 (3e-2
    against fp32 before).  Parameter gradients: BatchNorm's and InstanceNorm's backward subtract the common mode of a
    gradient map, which amplifies the rounding of the stored maps -- the emulation itself is 0.05-0.2 (relative L2) away
    from the fp32 gradient per tensor -- so every tensor of the device path must be no further from fp32 than 1.3 x the
    emulation is (+ 1e-2) and no further from the emulation than the emulation is from fp32 (+ 1e-2)."""
    ocfg, eng, p = _engine(L.BF16, B)
    x, y, eps = O.make_inputs(ocfg, B)
    info, grads, _ = O.train_step({k: v.clone() for k, v in p.items()}, ocfg, x, y, eps, {}, 11, 1)
    with O.bf16_storage():
        info16, grads16, _ = O.train_step({k: v.clone() for k, v in p.items()}, ocfg, x, y, eps, {}, 11, 1)
    x0 = eng.input_from_nchw(x.cuda())
    e = eng.eps_from_nchw(eps.cuda())
    eng.forward(x0, y.cuda(), e, B, True)
    eng.loss_sums()
    eng.loss_gate(11, 3.0, 0.01)
    eng.backward()
    s = eng.scal.cpu().numpy()
    for k in ("loss", "loss_rec", "loss_rec_p", "loss_p", "kld"):
        assert abs(s[L.SCAL[k]] - info16[k]) <= 2e-4 * abs(info16[k]), (k, s[L.SCAL[k]], float(info16[k]))
        assert abs(s[L.SCAL[k]] - info[k]) <= 2e-3 * abs(info[k]), (k, s[L.SCAL[k]], float(info[k]))
    live = [k for k in grads if not (k.endswith("1.1.bias") or k.endswith("2.0.bias"))]
    worst = (0.0, "")
    for k in live:
        dev = eng.grad[k].cpu()
        e_dev, e_emu, d = l2_err(dev, grads[k]), l2_err(grads16[k], grads[k]), l2_err(dev, grads16[k])
        assert e_dev <= 1.3 * e_emu + 1e-2, (k, e_dev, e_emu)
        assert d <= e_emu + 1e-2, (k, d, e_emu)
        worst = max(worst, (e_dev, k))
    a = torch.cat([eng.grad[k].flatten().cpu().double() for k in live])
    b = torch.cat([grads[k].flatten().double() for k in live])
    c = torch.cat([grads16[k].flatten().double() for k in live])
    cos = lambda u, v: float((u * v).sum() / u.norm() / v.norm())      # noqa: E731
    print(f"bf16 step, batch {B}: whole-gradient cosine device/fp32 {cos(a, b):.5f}, emulation/fp32 {cos(c, b):.5f}, "
          f"device/emulation {cos(a, c):.5f}; worst tensor vs fp32 {worst}")
    assert cos(a, b) > 0.998 and cos(a, c) > 0.998
    eng.adam_step()
    assert np.isfinite(float(eng.scal[L.SCAL["grad_norm"]]))


@pytest.mark.parametrize("stage", [0, 1])
def test_purification_bf16_close_to_fp32_and_shard_invariant(stage):
    ocfg, e32, p = _engine(L.F32, 256)
    _, e16, _ = _engine(L.BF16, 256)
    N = 512
    rng = np.random.default_rng(1)
    data = torch.from_numpy(rng.integers(0, 256, (N, 32, 32, 3), dtype=np.uint8)).cuda()
    labels = (torch.arange(N, device="cuda") % 10).long()
    outs = {}
    e32.inference_noise16 = True          # the noise stream of the bf16 engine (16-bit uniforms), so that both draw the same eps
    for name, eng in (("f32", e32), ("bf16", e16)):
        eng.seed = 5
        out = torch.zeros_like(data)
        noise = torch.zeros(N, 3, 32, 32, device="cuda")
        for lo in range(0, N, 256):
            eng.purify_batch(data, out, noise, labels, lo, 256, stage)
        outs[name] = (out, noise)
    d = (outs["f32"][0].int() - outs["bf16"][0].int()).abs()
    # the recipe's untrained decoder_p emits |p| ~ 1-4 (a trained one ~0.05), so bf16 rounding inside the decoder
    # is worth several LSB on a few pixels here
    assert int(d.max()) <= 16 and float((d <= 4).float().mean()) > 0.985 and float(d.float().mean()) < 0.75
    # bit-exact invariance to batching holds in bf16 as well
    out2 = torch.zeros_like(data)
    noise2 = torch.zeros(N, 3, 32, 32, device="cuda")
    for lo in range(0, N, 100):
        e16.purify_batch(data, out2, noise2, labels, lo, min(100, N - lo), stage)
    assert torch.equal(out2, outs["bf16"][0]) and torch.equal(noise2, outs["bf16"][1])
    # leaving out what main_vae_disentangle.py:61-62 never reads of the first pass changes no output bit
    out3 = torch.zeros_like(data)
    noise3 = torch.zeros(N, 3, 32, 32, device="cuda")
    for lo in range(0, N, 256):
        e16.purify_batch(data, out3, noise3, labels, lo, 256, stage, skip_unread=True)
    assert torch.equal(out3, outs["bf16"][0]) and torch.equal(noise3, outs["bf16"][1])


@pytest.mark.parametrize("batch", [64, 1000])
def test_one_launch_latent_path_equals_the_two_launches(batch):
    """fc_mu | fc_var + reparameterisation + post_conv (resnet.py:2245-2246, :2257-2259, :2197-2199) as one launch that
    never writes z, against the fused head followed by the post_conv launch: the same arithmetic in the same order, so
    post_conv's two outputs (z' and z' + class embedding, resnet.py:2212) and the purified bytes are equal bit for bit."""
    _, eng, _ = _engine(L.BF16, 1024)
    assert eng._fuse_head_post, "the one-launch latent path does not serve the CIFAR shape"
    N = 2000
    rng = np.random.default_rng(11)
    data = torch.from_numpy(rng.integers(0, 256, (N, 32, 32, 3), dtype=np.uint8)).cuda()
    labels = (torch.arange(N, device="cuda") % 10).long()
    eng.seed = 3
    res = []
    for fused in (True, False):
        eng._fuse_head_post = fused
        out = torch.zeros_like(data)
        noise = torch.zeros(N, 3, 32, 32, device="cuda")
        for lo in range(0, N, batch):
            eng.purify_batch(data, out, noise, labels, lo, min(batch, N - lo), 1)
        torch.cuda.synchronize()
        b = min(batch, N - (N - 1) // batch * batch)          # the last launch's images are what the buffers still hold
        res.append((out, noise, eng.act("z", eng.h, 128)[:b].clone(), eng.act("dec_p.in", eng.h, 128)[:b].clone()))
    eng._fuse_head_post = True
    assert int(res[0][0].max()) > 0 and torch.isfinite(res[0][1]).all()
    for got, want, what in zip(res[0], res[1], ("purified images", "perturbation", "post_conv output", "decoder_p input")):
        assert torch.equal(got, want), what


def test_purification_at_the_benchmark_batch_equals_small_batches():
    """Stage-1 purification (main_vae_disentangle.py:59-69) at the benchmark's 1024 images per launch -- every
    persistent kernel in steady state -- leaves exactly the bytes that 128-image launches leave: uint8 images and the
    recorded perturbation, bit for bit."""
    _, eng, _ = _engine(L.BF16, 1024)
    N = 2048
    rng = np.random.default_rng(7)
    data = torch.from_numpy(rng.integers(0, 256, (N, 32, 32, 3), dtype=np.uint8)).cuda()
    labels = (torch.arange(N, device="cuda") % 10).long()
    eng.seed = 9
    res = []
    for step in (1024, 128):
        out = torch.zeros_like(data)
        noise = torch.zeros(N, 3, 32, 32, device="cuda")
        for lo in range(0, N, step):
            eng.purify_batch(data, out, noise, labels, lo, step, 1)
        torch.cuda.synchronize()
        res.append((out, noise))
    assert int(res[0][0].max()) > 0 and torch.isfinite(res[0][1]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("B,cin,cout,hin", [(3, 64, 64, 32), (5, 128, 128, 16)])
def test_conv_bf16_fused_epilogue_and_accumulate(B, cin, cout, hin):
    """scale/shift + residual + ReLU epilogue and the accumulate mode: on the LDS-halo kernel (64 channels at 32x32) and
    on the image-resident kernel (128 channels at 16x16: layer2's second conv; its accumulate mode is the gather GEMM)."""
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    q = lambda t: t.bfloat16().float()      # noqa: E731
    w = q(torch.randn(cout, cin, 3, 3, generator=g) / 24)
    x = q(torch.randn(B, cin, hin, hin, generator=g))
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = q(torch.randn(B, cout, hin, hin, generator=g))
    conv = F.conv2d(x, w, None, 1, 1)
    ref = F.relu(q(conv * sc[None, :, None, None] + sh[None, :, None, None]) + res)
    d = L.ConvDesc(B, hin, hin, cin, cout, 3, 3, 1, 1, 0)
    st = L.stream()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), 0), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().data_ptr(), pk.data_ptr(), C.byref(d), 0, L.BF16, st))
    ya = torch.empty(B, hin, hin, cout, device="cuda", dtype=torch.bfloat16)
    scd, shd, resa = sc.cuda(), sh.cuda(), to_act(res, L.BF16)
    ep = L.Epilogue(scd.data_ptr(), shd.data_ptr(), resa.data_ptr(), 1, 0)
    L.check(lib.dvae_conv_fwd(to_act(x, L.BF16).data_ptr(), pk.data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.BF16, st))
    assert rel_err(from_act(ya, cout), ref) < 6e-3
    # dgrad with accumulate: dx <- dx + W^T dy
    pkd = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), 1), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().data_ptr(), pkd.data_ptr(), C.byref(d), 1, L.BF16, st))
    gy = q(torch.randn(B, cout, hin, hin, generator=g))
    base = q(torch.randn(B, cin, hin, hin, generator=g))
    dxa = to_act(base, L.BF16)
    L.check(lib.dvae_conv_dgrad(to_act(gy, L.BF16).data_ptr(), pkd.data_ptr(), dxa.data_ptr(), 1, C.byref(d), L.BF16, st))
    want = base + q(F.conv_transpose2d(gy, w, None, 1, 1))
    assert rel_err(from_act(dxa, cin), want) < 6e-3


PERSISTENT = [
    # cin, cout, k, stride, pad, transposed, hin, epilogue ("plain" | "res": scale/shift + residual + ReLU | "f32": fp32 out)
    (3, 64, 3, 1, 1, False, 32, "plain"),       # stem kernel
    (64, 64, 3, 1, 1, False, 32, "plain"),      # halo kernel
    (64, 64, 3, 1, 1, False, 32, "res"),        # halo kernel, residual + ReLU on the drain side
    (64, 32, 3, 1, 1, True, 32, "plain"),       # halo kernel, 32 output channels
    (32, 3, 3, 1, 1, True, 32, "f32"),          # halo kernel, fp32 output, two workgroups per CU
    (64, 128, 1, 2, 0, False, 32, "plain"),     # weights-resident 1x1, stride 2
    (256, 128, 1, 1, 0, False, 16, "plain"),    # weights-resident 1x1
    (128, 64, 4, 2, 1, True, 16, "plain"),      # image-resident transposed conv
    (128, 128, 3, 1, 1, False, 16, "res"),      # image-resident 3x3 with residual
    (64, 64, 3, 1, 1, False, 96, "res", 300),   # 36 tiles per image (not a power of two): whole-image walk, 300 images
    (64, 64, 3, 1, 1, False, 96, "plain", 60),  # ... fewer images than workgroups: tiles strided over the grid
    (32, 3, 3, 1, 1, True, 96, "f32", 120),
    (3, 64, 3, 1, 1, False, 96, "plain", 120),
    (128, 64, 4, 2, 1, True, 32, "plain", 300),   # tile-resident kernels: 4 tiles per image, 4.7 tiles per workgroup
    (128, 128, 3, 1, 1, False, 48, "res", 150),   # ... 9 tiles per image
    (128, 128, 3, 1, 1, False, 32, "res", 40),    # ... fewer tiles (160) than workgroups
    (64, 128, 3, 2, 1, False, 32, "plain"),       # stride-2 weights-in-registers kernel: 12.5 tiles per workgroup
    (64, 128, 3, 2, 1, False, 32, "bnrelu", 50),  # ... one tile per workgroup, scale / shift / ReLU
    (64, 128, 3, 2, 1, False, 96, "bnrelu", 130), # ... 36 tiles per image
    (32, 3, 3, 1, 1, True, 32, "f32", 50),        # decoder head kernel: fewer tiles than workgroups
    (128, 64, 4, 2, 1, True, 16, "bnrelu", 37),   # transposed-conv weights-in-registers kernel: 2 tiles per workgroup, scale / shift / ReLU
    (128, 64, 4, 2, 1, True, 16, "plain", 161),   # ... 6 tiles per workgroup, 4 in the last
    (128, 64, 4, 2, 1, True, 48, "bnrelu", 60),   # ... 72 tiles per image: three across
    (128, 128, 3, 1, 1, False, 16, "plain", 100), # 128 -> 128 weights-in-registers kernel without a residual: 1.6 tiles per workgroup
    (128, 128, 3, 1, 1, False, 16, "bnrelu", 321),  # ... 5.02 tiles per workgroup: the last one short
]


@pytest.mark.parametrize("hin,B", [(32, 5), (64, 3)])
def test_down_conv_epilogue(hin, B):
    """The stride-2 3x3 conv 64 -> 128 (resnet.py:26-31 with stride 2: layer2's first conv) with the folded BatchNorm + ReLU
    epilogue of the inference path, against torch on the bf16-rounded operands: <= 6e-3 of scale (one bf16 rounding)."""
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    q = lambda t: t.bfloat16().float()      # noqa: E731
    w = q(torch.randn(128, 64, 3, 3, generator=g) / 24.0)
    x = q(torch.randn(B, 64, hin, hin, generator=g))
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    want = torch.relu(F.conv2d(x, w, None, stride=2, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    d = L.ConvDesc(B, hin, hin, 64, 128, 3, 3, 2, 1, 0)
    st = L.stream()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    ya = torch.full((B, hin // 2, hin // 2, 128), 7.0, device="cuda", dtype=torch.bfloat16)
    scd, shd = sc.cuda(), sh.cuda()
    ep = L.Epilogue(scd.data_ptr(), shd.data_ptr(), None, 1, 0)
    L.check(lib.dvae_conv_fwd(to_act(x, L.BF16).data_ptr(), pk.data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.BF16, st))
    assert rel_err(from_act(ya, 128), want) < 6e-3
    # the gather GEMM on the same launch (the dedicated kernel switched off) rounds the same fp32 sums: equal up to the
    # order of summation
    yb = torch.full_like(ya, 7.0)
    L.check(lib.dvae_set_option(b"conv_down", 0))
    try:
        L.check(lib.dvae_conv_fwd(to_act(x, L.BF16).data_ptr(), pk.data_ptr(), C.byref(ep), yb.data_ptr(), C.byref(d), L.BF16, st))
    finally:
        L.check(lib.dvae_set_option(b"conv_down", 1))
    assert rel_err(ya.float().cpu(), yb.float().cpu()) < 4e-3


@pytest.mark.parametrize("hin,B", [(16, 1), (16, 5), (16, 64), (32, 3), (48, 2), (112, 1)])
def test_up_conv_epilogue(hin, B):
    """ConvTranspose2d k4 s2 p1 128 -> 64 on 16x16 maps (resnet.py:2139-2167: the decoders' first layer; 32x32 / 112x112 maps
    at 64x64 / 224x224 images, 48x48 for a tile count that is no power of two) with a per-channel
    scale / shift + ReLU epilogue, against torch on the bf16-rounded operands: <= 6e-3 of scale (one bf16 rounding); and
    against the image-resident kernel on the same launch (the dedicated kernel switched off): the same fp32 sums in the
    same order, so the same bits."""
    lib = L.lib()
    g = torch.Generator().manual_seed(9)
    q = lambda t: t.bfloat16().float()      # noqa: E731
    w = q(torch.randn(128, 64, 4, 4, generator=g) / 32.0)
    x = q(torch.randn(B, 128, hin, hin, generator=g))
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    want = torch.relu(F.conv_transpose2d(x, w, None, stride=2, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    d = L.ConvDesc(B, hin, hin, 128, 64, 4, 4, 2, 1, 1)
    st = L.stream()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    ya = torch.full((B, 2 * hin, 2 * hin, 64), 7.0, device="cuda", dtype=torch.bfloat16)
    scd, shd = sc.cuda(), sh.cuda()
    ep = L.Epilogue(scd.data_ptr(), shd.data_ptr(), None, 1, 0)
    xa = to_act(x, L.BF16)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk.data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.BF16, st))
    assert rel_err(from_act(ya, 64), want) < 6e-3
    yb = torch.full_like(ya, 7.0)
    L.check(lib.dvae_set_option(b"conv_up", 0))
    try:
        L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk.data_ptr(), C.byref(ep), yb.data_ptr(), C.byref(d), L.BF16, st))
    finally:
        L.check(lib.dvae_set_option(b"conv_up", 1))
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)


@pytest.mark.parametrize("hin,B,res", [(16, 1, True), (16, 7, False), (16, 64, True), (32, 3, True), (48, 2, True), (112, 1, False)])
def test_mid_conv_epilogue(hin, B, res):
    """Conv2d 3x3 128 -> 128 (resnet.py:26-31: a BasicBlock's second conv in layer2) with the folded BatchNorm + shortcut +
    ReLU epilogue of the inference path (resnet.py:43-47), against torch on the bf16-rounded operands: <= 6e-3 of scale
    (one bf16 rounding); and against the image-resident kernel on the same launch (the dedicated kernel switched off):
    the same fp32 sums in the same order, so the same bits."""
    lib = L.lib()
    g = torch.Generator().manual_seed(13)
    q = lambda t: t.bfloat16().float()      # noqa: E731
    w = q(torch.randn(128, 128, 3, 3, generator=g) / 34.0)
    x = q(torch.randn(B, 128, hin, hin, generator=g))
    r = q(torch.randn(B, 128, hin, hin, generator=g)) if res else None
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    want = F.conv2d(x, w, None, stride=1, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    want = torch.relu(want + r if res else want)
    d = L.ConvDesc(B, hin, hin, 128, 128, 3, 3, 1, 1, 0)
    st = L.stream()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    ya = torch.full((B, hin, hin, 128), 7.0, device="cuda", dtype=torch.bfloat16)
    scd, shd = sc.cuda(), sh.cuda()
    ra = to_act(r, L.BF16) if res else None
    ep = L.Epilogue(scd.data_ptr(), shd.data_ptr(), None if ra is None else ra.data_ptr(), 1, 0)
    xa = to_act(x, L.BF16)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk.data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.BF16, st))
    assert rel_err(from_act(ya, 128), want) < 6e-3
    yb = torch.full_like(ya, 7.0)
    L.check(lib.dvae_set_option(b"conv_mid", 0))
    try:
        L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk.data_ptr(), C.byref(ep), yb.data_ptr(), C.byref(d), L.BF16, st))
    finally:
        L.check(lib.dvae_set_option(b"conv_mid", 1))
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)


@pytest.mark.parametrize("cfg", PERSISTENT, ids=lambda c: "-".join(str(v) for v in c))
def test_persistent_kernels_do_not_depend_on_the_batch(cfg):
    """The specialised conv kernels are persistent: a workgroup walks many tiles (or images) of a large launch, with
    the next tile prefetched and the previous one drained inside the current tile's loop.  The small-batch parity
    tests give every workgroup at most one tile, so this one runs 800 images (3 to 13 tiles per workgroup) and
    requires every image's output to equal, bit for bit, what a 3-image launch computes for it: a conv has no
    cross-image term (resnet.py:26-31, :2139-2167), so any difference is a tile-loop hazard."""
    cin, cout, k, stride, pad, tr, hin, mode = cfg[:8]
    lib = L.lib()
    B, b = (cfg[8] if len(cfg) > 8 else 800), 3
    g = torch.Generator().manual_seed(17 + cin + cout)
    sp = ConvSpec("t", cin, cout, k, stride, pad, tr, hin, True)
    w = torch.randn(sp.wshape, generator=g) / (cin * k * k) ** 0.5
    ho = (hin - 1) * stride - 2 * pad + k if tr else (hin + 2 * pad - k) // stride + 1
    f32out = mode == "f32"
    cs_in, cs_out = L.act_channels(cin, L.BF16), L.act_channels(cout, L.F32 if f32out else L.BF16)
    x = torch.randn(B, hin, hin, cs_in, generator=g).cuda().bfloat16()
    if cs_in > cin:
        x[..., cin:] = 0
    res = torch.randn(B, ho, ho, cs_out, generator=g).cuda().bfloat16() if mode == "res" else None
    sc = (torch.rand(cout, generator=g) + 0.5).cuda() if mode in ("res", "bnrelu") else None
    sh = torch.randn(cout, generator=g).cuda()
    st = L.stream()
    odt = torch.float32 if f32out else torch.bfloat16

    def run(lo, n):
        d = L.ConvDesc(n, hin, hin, cin, cout, k, k, stride, pad, int(tr))
        pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
        y = torch.full((n, ho, ho, cs_out), 3.0, device="cuda", dtype=odt)
        ep = L.Epilogue(L.ptr(sc), sh.data_ptr(), None if res is None else res[lo:lo + n].data_ptr(), int(mode in ("res", "bnrelu")),
                        int(f32out))
        L.check(lib.dvae_conv_fwd(x[lo:lo + n].data_ptr(), pk.data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))
        torch.cuda.synchronize()
        return y

    big = run(0, B)
    assert torch.isfinite(big.float()).all()
    for lo in (0, B // 2 - 2, B - b):
        small = run(lo, b)
        assert torch.equal(big[lo:lo + b, ..., :cout], small[..., :cout]), f"images {lo}..{lo + b - 1} differ"


BACKWARD = [
    (64, 64, 3, 1, 1, False, 32),       # all-tap weight gradient; halo kernel as input-gradient
    (64, 32, 3, 1, 1, True, 32),        # all-tap weight gradient, 32 x 64 stored channels: two pixel-row groups per workgroup
    (32, 3, 3, 1, 1, True, 32),         # ... 8 x 32: four row groups
    (3, 64, 3, 1, 1, False, 32),        # ... 8 x 64 (the first conv)
    (64, 128, 3, 2, 1, False, 32),      # slab weight gradient; image-resident input-gradient (9 slabs)
    (128, 64, 4, 2, 1, True, 16),
    (128, 128, 3, 1, 1, False, 16),
    # 512 images of 16 x 16 maps = 131 072 pixel rows: past the 100 000 rows from which the wide slabs go to the
    # large-tile weight-gradient kernel (256 x 128 tiles, both bias-partial modes, the 1152 = 4.5 x 256 K tail)
    (64, 128, 3, 2, 1, False, 32, 512),
    (128, 64, 4, 2, 1, True, 16, 512),
    (128, 128, 3, 1, 1, False, 16, 512),
    (64, 64, 3, 1, 1, False, 96, 256),  # 36 tiles per image: not a power of two
    (64, 128, 3, 2, 1, False, 64, 256), # tile-resident input-gradient over 32x32 gradient maps (4 tiles per image)
]


@pytest.mark.parametrize("cfg", BACKWARD, ids=lambda c: "-".join(str(v) for v in c))
def test_backward_kernels_at_train_batch(cfg):
    """Input- and weight-gradient kernels at the reference's batch of 128 per launch (configs/cifar10/resnet18DVAE.yaml;
    384 here: three such batches), where their workgroups loop over many tiles: the input gradient of an image does
    not depend on the batch (bit-exact against 3-image launches), and the weight / bias gradient is linear in the
    batch (the 384-image launch against the sum of three 128-image launches, fp32 summation order apart: 1e-4)."""
    cin, cout, k, stride, pad, tr, hin = cfg[:7]
    lib = L.lib()
    B, b = (cfg[7] if len(cfg) > 7 else 384), 3
    g = torch.Generator().manual_seed(29 + cin + cout + k)
    sp = ConvSpec("t", cin, cout, k, stride, pad, tr, hin, True)
    w = torch.randn(sp.wshape, generator=g) / (cin * k * k) ** 0.5
    ho = (hin - 1) * stride - 2 * pad + k if tr else (hin + 2 * pad - k) // stride + 1
    cs_in, cs_out = L.act_channels(cin, L.BF16), L.act_channels(cout, L.BF16)
    x = torch.randn(B, hin, hin, cs_in, generator=g).cuda().bfloat16()
    gy = torch.randn(B, ho, ho, cs_out, generator=g).cuda().bfloat16()
    st = L.stream()

    def dgrad(lo, n):
        d = L.ConvDesc(n, hin, hin, cin, cout, k, k, stride, pad, int(tr))
        pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_DGRAD), device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_DGRAD, L.BF16, st))
        dx = torch.full((n, hin, hin, cs_in), 3.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_conv_dgrad(gy[lo:lo + n].data_ptr(), pk.data_ptr(), dx.data_ptr(), 0, C.byref(d), L.BF16, st))
        torch.cuda.synchronize()
        return dx

    def wgrad(lo, n):
        d = L.ConvDesc(n, hin, hin, cin, cout, k, k, stride, pad, int(tr))
        ws = torch.empty(lib.dvae_conv_wgrad_workspace_bytes(C.byref(d)) // 4 + 4, device="cuda")
        dw = torch.full(sp.wshape, 9.0, device="cuda")
        db = torch.full((cout,), 9.0, device="cuda")
        L.check(lib.dvae_conv_wgrad(x[lo:lo + n].data_ptr(), gy[lo:lo + n].data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                    ws.numel() * 4, C.byref(d), L.BF16, st))
        torch.cuda.synchronize()
        return dw, db

    big = dgrad(0, B)
    assert torch.isfinite(big.float()).all()
    for lo in (0, 190, B - b):
        assert torch.equal(big[lo:lo + b, ..., :cin], dgrad(lo, b)[..., :cin]), f"input gradient of images {lo}.. differs"
    dw, db = wgrad(0, B)
    if B > 384:
        d = L.ConvDesc(B, hin, hin, cin, cout, k, k, stride, pad, int(tr))
        assert lib.dvae_conv_wgrad_uses_big_tiles(C.byref(d), L.BF16) == 1
    parts = [wgrad(lo, 128) for lo in range(0, B, 128)]
    assert rel_err(dw.cpu(), sum(p_[0] for p_ in parts).cpu()) < 1e-4
    assert rel_err(db.cpu(), sum(p_[1] for p_ in parts).cpu()) < 1e-4


@pytest.mark.parametrize("h,B", [(16, 800), (32, 300), (48, 100)])
def test_fused_encoder_head_does_not_depend_on_the_batch(h, B):
    """The fused head walks images (16x16 tiles of larger maps) with a persistent workgroup too, and its noise is keyed
    by the global image index and the element's place in the image (resnet.py:2257-2259 draws per element): image i of
    a large launch equals image i of a 3-image launch that starts its numbering at i, bit for bit."""
    lib = L.lib()
    cin, Ld, b = 128, 256, 3
    g = torch.Generator().manual_seed(23)
    w = torch.randn(2 * Ld, cin, 1, 1, generator=g) / cin ** 0.5
    bias = (0.1 * torch.randn(2 * Ld, generator=g)).cuda()
    x = torch.randn(B, h, h, cin, generator=g).cuda().bfloat16()
    st = L.stream()

    def run(lo, n):
        d = L.ConvDesc(n, h, h, cin, 2 * Ld, 1, 1, 1, 0, 0)
        pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
        z = torch.full((n, h, h, Ld), 7.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_fc_reparam_fwd_philox(x[lo:lo + n].data_ptr(), pk.data_ptr(), bias.data_ptr(), z.data_ptr(), C.byref(d),
                                               Ld, None, 1000 + lo, 0xfeedbeef12345, 2, L.BF16, st))
        torch.cuda.synchronize()
        return z

    big = run(0, B)
    assert torch.isfinite(big.float()).all()
    for lo in (0, B // 2 + 11, B - b):
        assert torch.equal(big[lo:lo + b], run(lo, b)), f"images {lo}..{lo + b - 1} differ"


@pytest.mark.parametrize("B,Ld,h,stream_id", [(3, 256, 16, 5), (5, 128, 16, 5), (3, 256, 16, 0x80000005), (5, 128, 16, 0x80000005),
                                              (3, 256, 32, 0x80000005), (2, 256, 24, 5), (3, 128, 32, 5), (2, 256, 48, 0x80000005),
                                              (2, 128, 112, 5)])
def test_fused_encoder_head_bf16(B, Ld, h, stream_id):
    """fc_mu|fc_var (1x1 conv, resnet.py:2245-2246) + reparameterise (resnet.py:2257-2259) in one launch against
    torch fp32 on the same bf16-rounded operands and the same Philox noise; only the stored z is rounded to bf16.
    Maps whose sides are multiples of 16 run on the tile-resident kernel, other sizes on the gather GEMM; stream ids with bit 31 draw eight
    normals per Philox call (the two lane halves of a pixel trade halves of each group): the in-kernel draw must equal what
    dvae_philox_normal materialises for the same key."""
    lib = L.lib()
    cin = 128
    g = torch.Generator().manual_seed(11 + Ld)
    q = lambda t: t.bfloat16().float()      # noqa: E731
    w = q(torch.randn(2 * Ld, cin, 1, 1, generator=g) / cin ** 0.5)
    bias = 0.1 * torch.randn(2 * Ld, generator=g)
    x = q(torch.randn(B, cin, h, h, generator=g))
    mulv = F.conv2d(x, w, bias)
    d = L.ConvDesc(B, h, h, cin, 2 * Ld, 1, 1, 1, 0, 0)
    assert lib.dvae_fc_reparam_supported(C.byref(d), Ld, L.BF16) == 1
    assert lib.dvae_fc_reparam_supported(C.byref(d), Ld, L.F32) == 0
    st = L.stream()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    index = torch.tensor([7, 123456, 3, 49999, 0][:B], device="cuda", dtype=torch.int32)
    seed = 0x1234567890abcdef
    eps = torch.empty(B, h, h, Ld, device="cuda")
    L.check(lib.dvae_philox_normal(eps.data_ptr(), index.data_ptr(), 0, B, h * h * Ld, seed, stream_id, st))
    z_ref = eps.cpu().permute(0, 3, 1, 2) * torch.exp(0.5 * mulv[:, Ld:]) + mulv[:, :Ld]
    xa = to_act(x, L.BF16)
    z = torch.full((B, h, h, Ld), 7.0, device="cuda", dtype=torch.bfloat16)
    bdev = bias.cuda()
    L.check(lib.dvae_fc_reparam_fwd_philox(xa.data_ptr(), pk.data_ptr(), bdev.data_ptr(), z.data_ptr(), C.byref(d), Ld,
                                           index.data_ptr(), 0, seed, stream_id, L.BF16, st))
    assert rel_err(from_act(z, Ld), z_ref) < 6e-3
    # contiguous image numbering (index NULL, first image given) draws the same stream
    L.check(lib.dvae_philox_normal(eps.data_ptr(), None, 40000, B, h * h * Ld, seed, stream_id, st))
    z_ref = eps.cpu().permute(0, 3, 1, 2) * torch.exp(0.5 * mulv[:, Ld:]) + mulv[:, :Ld]
    L.check(lib.dvae_fc_reparam_fwd_philox(xa.data_ptr(), pk.data_ptr(), bdev.data_ptr(), z.data_ptr(), C.byref(d), Ld,
                                           None, 40000, seed, stream_id, L.BF16, st))
    assert rel_err(from_act(z, Ld), z_ref) < 6e-3


@pytest.mark.parametrize("H,W", [(32, 48), (48, 16), (16, 80)])
def test_tile_resident_kernels_on_non_square_maps(H, W):
    """The three flavours of the tile-resident kernel (transposed conv, 3x3 128 -> 128 with the residual epilogue, fused
    encoder head) on H x W maps with H != W: rows and columns of the tile grid must not be confused anywhere."""
    lib = L.lib()
    B, cin = 3, 128
    g = torch.Generator().manual_seed(H * 100 + W)
    q = lambda t: t.bfloat16().float()      # noqa: E731
    st = L.stream()
    x = q(torch.randn(B, cin, H, W, generator=g))
    xa = to_act(x, L.BF16)

    def packed(d, w, op=L.OP_FWD):
        pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), op), device="cuda", dtype=torch.bfloat16)
        L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), op, L.BF16, st))
        return pk

    # ConvTranspose2d(128, 64, 4, 2, 1) + bias + ReLU
    w = q(torch.randn(cin, 64, 4, 4, generator=g) / (cin * 4) ** 0.5)
    bias = torch.randn(64, generator=g)
    d = L.ConvDesc(B, H, W, cin, 64, 4, 4, 2, 1, 1)
    y = torch.full((B, 2 * H, 2 * W, 64), 7.0, device="cuda", dtype=torch.bfloat16)
    bd = bias.cuda()
    ep = L.Epilogue(None, bd.data_ptr(), None, 1, 0)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), packed(d, w).data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))
    assert rel_err(from_act(y, 64), F.relu(F.conv_transpose2d(x, w, bias, stride=2, padding=1))) < 6e-3
    # Conv2d(128, 128, 3, 1, 1) * scale + shift + residual, ReLU
    w = q(torch.randn(128, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    res = q(torch.randn(B, 128, H, W, generator=g))
    d = L.ConvDesc(B, H, W, cin, 128, 3, 3, 1, 1, 0)
    y = torch.full((B, H, W, 128), 7.0, device="cuda", dtype=torch.bfloat16)
    scd, shd, resa = sc.cuda(), sh.cuda(), to_act(res, L.BF16)
    ep = L.Epilogue(scd.data_ptr(), shd.data_ptr(), resa.data_ptr(), 1, 0)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), packed(d, w).data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))
    ref = F.relu(q(F.conv2d(x, w, None, 1, 1) * sc[None, :, None, None] + sh[None, :, None, None]) + res)
    assert rel_err(from_act(y, 128), ref) < 6e-3
    # input-gradient of Conv2d(64, 128, 3, 2, 1) from an H x W gradient map
    w = q(torch.randn(128, 64, 3, 3, generator=g) / 24)
    d = L.ConvDesc(B, 2 * H, 2 * W, 64, 128, 3, 3, 2, 1, 0)
    dx = torch.full((B, 2 * H, 2 * W, 64), 7.0, device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_dgrad(xa.data_ptr(), packed(d, w, L.OP_DGRAD).data_ptr(), dx.data_ptr(), 0, C.byref(d), L.BF16, st))
    assert rel_err(from_act(dx, 64), F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1)) < 6e-3
    # fused encoder head
    Ld, seed, stream_id = 128, 99, 0x80000003
    w = q(torch.randn(2 * Ld, cin, 1, 1, generator=g) / cin ** 0.5)
    bias = (0.1 * torch.randn(2 * Ld, generator=g))
    mulv = F.conv2d(x, w, bias)
    d = L.ConvDesc(B, H, W, cin, 2 * Ld, 1, 1, 1, 0, 0)
    eps = torch.empty(B, H, W, Ld, device="cuda")
    L.check(lib.dvae_philox_normal(eps.data_ptr(), None, 500, B, H * W * Ld, seed, stream_id, st))
    z = torch.full((B, H, W, Ld), 7.0, device="cuda", dtype=torch.bfloat16)
    bd = bias.cuda()
    L.check(lib.dvae_fc_reparam_fwd_philox(xa.data_ptr(), packed(d, w).data_ptr(), bd.data_ptr(), z.data_ptr(), C.byref(d), Ld,
                                           None, 500, seed, stream_id, L.BF16, st))
    assert rel_err(from_act(z, Ld), eps.cpu().permute(0, 3, 1, 2) * torch.exp(0.5 * mulv[:, Ld:]) + mulv[:, :Ld]) < 6e-3


@pytest.mark.parametrize("cin,cout,hin,tr,B", [(64, 64, 32, True, 5), (64, 32, 32, True, 5), (32, 64, 64, False, 5),
                                               (64, 32, 32, True, 600), (64, 64, 96, True, 5), (64, 32, 96, True, 300)])
def test_conv_output_statistics_bf16(cin, cout, hin, tr, B):
    """InstanceNorm2d statistics (resnet.py:2030-2046) gathered by the producing conv's store against the
    stand-alone statistics kernel on the stored tensor: same values up to fp32 summation order.  With 600 images a
    workgroup walks two or three of them (the sums are reset and flushed inside its tile loop)."""
    lib = L.lib()
    g = torch.Generator().manual_seed(3 + cout + hin)
    sp = ConvSpec("t", cin, cout, 3, 1, 1, tr, hin, True)
    w = torch.randn(sp.wshape, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g).cuda()
    x = torch.randn(B, cin, hin, hin, generator=g) + 0.5
    d = L.ConvDesc(B, hin, hin, cin, cout, 3, 3, 1, 1, int(tr))
    st = L.stream()
    n = lib.dvae_conv_out_stats_floats(C.byref(d), L.BF16)
    cs = L.act_channels(cout, L.BF16)
    assert n == B * 8 * 2 * cs
    assert lib.dvae_conv_out_stats_floats(C.byref(d), L.F32) == 0
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    xa = to_act(x, L.BF16)
    ya = torch.empty(B, hin, hin, cs, device="cuda", dtype=torch.bfloat16)
    chunks = torch.full((n,), float("nan"), device="cuda")
    ep = L.Epilogue(None, L.ptr(bias), None, 0, 0, None, 0, chunks.data_ptr())
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk.data_ptr(), C.byref(ep), ya.data_ptr(), C.byref(d), L.BF16, st))
    out = {}
    for tag in ("fused", "alone"):
        out[tag] = [torch.empty(B, cs, device="cuda"), torch.empty(B, cs, device="cuda"), torch.empty(B, 2, cs, device="cuda")]
    m, i, s = out["fused"]
    L.check(lib.dvae_norm_stats_from_chunks(chunks.data_ptr(), hin * hin // 8, None, None, m.data_ptr(), i.data_ptr(),
                                            s.data_ptr(), B, hin * hin, cs, 1e-5, L.BF16, st))
    m, i, s = out["alone"]
    ws = torch.empty(lib.dvae_norm_workspace_bytes(B, hin * hin, cs) // 4 + 4, device="cuda")
    L.check(lib.dvae_norm_fwd_stats(ya.data_ptr(), None, None, m.data_ptr(), i.data_ptr(), s.data_ptr(), ws.data_ptr(),
                                    ws.numel() * 4, B, hin * hin, cs, 1e-5, L.BF16, st))
    for a_, b_ in zip(out["fused"], out["alone"]):
        assert torch.isfinite(a_[..., :cout]).all()
        assert rel_err(a_[..., :cout], b_[..., :cout]) < 2e-5
    # and against torch on the stored values
    y = from_act(ya, cout).float()
    assert rel_err(out["fused"][0][:, :cout].cpu(), y.mean(dim=(2, 3))) < 2e-5


@pytest.mark.parametrize("cout,res,stats", [(64, False, False), (64, True, False), (64, False, True), (32, False, True)])
def test_reverse_walk_changes_no_bit(cout, res, stats):
    """dvae_epilogue.reverse_walk is a scheduling hint: the weights-in-registers 3x3 conv (resnet.py:26-31, :2139-2167)
    sweeping a 700-image batch last image to first -- 2.7 images per workgroup, so the last workgroups have one image
    fewer -- stores the same bytes, and the same output statistics, as sweeping it first to last."""
    lib = L.lib()
    B, hin, cin = 700, 32, 64
    g = torch.Generator().manual_seed(23 + cout)
    w = torch.randn(cout, cin, 3, 3, generator=g) / 24.0
    x = torch.randn(B, hin, hin, cin, generator=g).cuda().bfloat16()
    r = torch.randn(B, hin, hin, cout, generator=g).cuda().bfloat16() if res else None
    sh = torch.randn(cout, generator=g).cuda()
    d = L.ConvDesc(B, hin, hin, cin, cout, 3, 3, 1, 1, 0)
    st = L.stream()
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    n = lib.dvae_conv_out_stats_floats(C.byref(d), L.BF16) if stats else 0
    got = []
    for rev in (0, 1):
        y = torch.full((B, hin, hin, cout), 5.0, device="cuda", dtype=torch.bfloat16)
        ch = torch.full((max(n, 1),), float("nan"), device="cuda")
        ep = L.Epilogue(None, sh.data_ptr(), L.ptr(r), int(res), 0, None, 0, ch.data_ptr() if stats else None, None, None, None, 0, rev)
        L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), y.data_ptr(), C.byref(d), L.BF16, st))
        torch.cuda.synchronize()
        got.append((y, ch))
    assert torch.isfinite(got[0][0].float()).all()
    assert torch.equal(got[0][0], got[1][0])
    if stats:
        assert torch.isfinite(got[0][1]).all() and torch.equal(got[0][1], got[1][1])


@pytest.mark.parametrize("cin,cout,B,f32out,hin", [(64, 32, 3, False, 32), (64, 32, 600, False, 32), (32, 3, 700, True, 32),
                                                   (64, 32, 300, False, 96), (32, 3, 300, True, 96)])
def test_conv_input_normalisation_prologue_bf16(cin, cout, B, f32out, hin):
    """The decoders' convs consume ReLU(InstanceNorm2d(x)) (resnet.py:2030-2046): the per-image scale / shift and the
    ReLU applied while the conv stages its input, against the same conv on a tensor normalised beforehand.  600 / 700
    images: more than the launch has workgroups, so a workgroup changes image (and scale / shift) inside its loop."""
    lib = L.lib()
    g = torch.Generator().manual_seed(11 + cout + B)
    sp = ConvSpec("t", cin, cout, 3, 1, 1, True, hin, True)
    w = torch.randn(sp.wshape, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g).cuda()
    d = L.ConvDesc(B, hin, hin, cin, cout, 3, 3, 1, 1, 1)
    assert lib.dvae_conv_supports_input_norm(C.byref(d), L.BF16) == 1
    st = L.stream()
    cs_in, cs_out = L.act_channels(cin, L.BF16), L.act_channels(cout, L.F32 if f32out else L.BF16)
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), L.OP_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().contiguous().data_ptr(), pk.data_ptr(), C.byref(d), L.OP_FWD, L.BF16, st))
    xa = (torch.randn(B, hin, hin, cs_in, generator=g) * 1.5).cuda().bfloat16()
    ss = torch.empty(B, 2, cs_in)
    ss[:, 0] = torch.rand(B, cs_in, generator=g) + 0.5
    ss[:, 1] = torch.randn(B, cs_in, generator=g) * 0.5
    ss = ss.cuda()
    odt = torch.float32 if f32out else torch.bfloat16
    y_fused = torch.empty(B, hin, hin, cs_out, device="cuda", dtype=odt)
    y_plain = torch.empty_like(y_fused)
    ep = L.Epilogue(None, L.ptr(bias), None, 0, int(f32out), ss.data_ptr(), 1)
    L.check(lib.dvae_conv_fwd(xa.data_ptr(), pk.data_ptr(), C.byref(ep), y_fused.data_ptr(), C.byref(d), L.BF16, st))
    xn = torch.relu(xa.float() * ss[:, 0, None, None, :] + ss[:, 1, None, None, :]).bfloat16()
    ep = L.Epilogue(None, L.ptr(bias), None, 0, int(f32out))
    L.check(lib.dvae_conv_fwd(xn.data_ptr(), pk.data_ptr(), C.byref(ep), y_plain.data_ptr(), C.byref(d), L.BF16, st))
    torch.cuda.synchronize()
    # the prologue may fuse the multiply-add (one rounding less before the bf16 rounding of the staged value): a few
    # staged values differ by one bf16 step, and a bf16 output may then round the other way (one step = 4e-3 of a value)
    assert rel_err(y_fused[..., :cout].float().cpu(), y_plain[..., :cout].float().cpu()) < (1e-3 if f32out else 6e-3)
    # per image: an image normalised with another image's scale / shift would be off by O(1)
    per_img = (y_fused[..., :cout].float() - y_plain[..., :cout].float()).abs().amax(dim=(1, 2, 3))
    scale = y_plain[..., :cout].float().abs().amax()
    assert (per_img < 0.05 * scale).all()


@pytest.mark.parametrize("dtype", [1, 0], ids=["bf16", "f32"])
def test_graphed_train_steps_equal_eager_steps(dtype):
    """Four steps replayed from captured graphs leave exactly the parameters, Adam moments, BatchNorm buffers and
    loss scalars that four launch-by-launch steps leave (no kernel on the path uses atomics, so runs are repeatable)."""
    B = 8
    ocfg = O.DVAEConfig()
    g = torch.Generator().manual_seed(5)
    xs = torch.rand(4, B, 3, 32, 32, generator=g).cuda()
    ys = torch.randint(0, 10, (4, B), generator=g).cuda()
    ends = []
    for graphed in (False, True):
        _, eng, _ = _engine(dtype, B)
        eng.set_hyper(lr=5e-3, max_norm=5.0, grad_scale=1.0, steps_done=0)
        scal = []
        for i in range(4):
            x0 = eng.input_from_nchw(xs[i])
            eps = eng.eps_philox(B, first=i * B, stream_id=12)
            step = eng.train_step_graphed if graphed else eng.train_step
            step(x0, ys[i], eps, B, 11, 3.0, 0.01)
            scal.append(eng.scal.clone())
        torch.cuda.synchronize()
        ends.append((eng.P.clone(), eng.M.clone(), eng.V.clone(), eng.R.clone(), torch.stack(scal), eng.hyper.clone()))
    for a, b in zip(*ends):
        assert torch.equal(a, b)
    assert float(ends[0][5][6]) == 4.0          # the Adam step count advanced on the device in both modes


@pytest.mark.parametrize("B,hw,c", [(5, 32, 64), (3, 32, 32), (2, 16, 64)])
def test_instance_norm_per_image_kernels_bf16(B, hw, c):
    """InstanceNorm+ReLU of images that fit one workgroup's registers (one launch each way) against torch autograd
    in fp32 on the same bf16-rounded inputs; an image's result must not depend on the rest of the batch."""
    lib = L.lib()
    g = torch.Generator().manual_seed(c + B)
    x = (torch.randn(B, c, hw, hw, generator=g) * 1.5 + 0.3).bfloat16().float()
    dy = torch.randn(B, c, hw, hw, generator=g).bfloat16().float()
    xr = x.clone().requires_grad_(True)
    yr = F.relu(F.instance_norm(xr, eps=1e-5))
    yr.backward(dy)
    xa, dya = to_act(x, L.BF16), to_act(dy, L.BF16)
    ya, dxa = torch.empty_like(xa), torch.empty_like(xa)
    mean = torch.empty(B, c, device="cuda")
    invstd = torch.empty(B, c, device="cuda")
    ws = torch.empty(lib.dvae_norm_workspace_bytes(B, hw * hw, c) // 4 + 16, device="cuda")
    st = L.stream()

    def run(n):
        L.check(lib.dvae_norm_fwd_train(L.ptr(xa), None, L.ptr(ya), None, None, None, None, L.ptr(mean), L.ptr(invstd), L.ptr(ws),
                                        ws.numel() * 4, n, hw * hw, c, 1, 0.0, 1e-5, L.BF16, st))
        L.check(lib.dvae_norm_bwd(L.ptr(dya), L.ptr(ya), L.ptr(xa), L.ptr(dxa), None, None, L.ptr(mean), L.ptr(invstd), None, None,
                                  L.ptr(ws), ws.numel() * 4, n, hw * hw, c, 1, L.BF16, st))
    run(B)
    y, dx = from_act(ya, c), from_act(dxa, c)
    assert rel_err(y, yr.detach()) < 6e-3
    assert rel_err(dx, xr.grad) < 8e-3
    assert rel_err(mean.cpu(), x.mean(dim=(2, 3))) < 1e-5
    y_all, dx_all = ya.clone(), dxa.clone()
    ya.zero_(); dxa.zero_()
    run(2)                                   # the first two images alone
    assert torch.equal(ya[:2], y_all[:2]) and torch.equal(dxa[:2], dx_all[:2])


@pytest.mark.parametrize("train", [True, False])
def test_model_forward_bf16_64x64(train):
    """Tiny-ImageNet shape (64x64, 200 classes, plain class embedding): the tile loops see two tiles per row, the
    per-image InstanceNorm kernels and the image-resident conv do not apply and the generic kernels take over."""
    kw = dict(image_size=64, spatial_emb=False, num_classes=200)
    ocfg, cfg = O.DVAEConfig(**kw), DVAEConfig(**kw)
    eng = DVAEEngine(cfg, 2, "cuda", L.BF16)
    p = O.recipe_params(ocfg)
    eng.load_params(p)
    x, y, eps = O.make_inputs(ocfg, 2)
    with torch.no_grad():
        ref = O.forward(p, ocfg, x, y, eps, train, {})
    x0 = eng.input_from_nchw(x.cuda())
    e = eng.eps_from_nchw(eps.cuda())
    rec_pre, pp = eng.forward(x0, y.cuda(), e, 2, train)
    rec = eng.loss_sums()
    assert l2_err(from_act(rec[:2], 3), ref["rec"]) < 3e-2
    assert l2_err(from_act(pp[:2], 3), ref["perturbation"]) < 6e-2
    if train:
        eng.loss_gate(11, 3.0, 0.01)
        eng.backward()
        eng.adam_step()
        torch.cuda.synchronize()
        assert np.isfinite(float(eng.scal[L.SCAL["grad_norm"]]))


def test_post_conv_second_output_equals_embedding_add():
    """post_conv with the fused second output (z + class embedding, decoder_p's input) against the conv followed by the
    stand-alone embedding add: same bits for both outputs."""
    lib = L.lib()
    B, hw, ncls = 6, 16, 10
    g = torch.Generator().manual_seed(9)
    d = L.ConvDesc(B, hw, hw, 256, 128, 1, 1, 1, 0, 0)
    assert lib.dvae_conv_supports_emb_out(C.byref(d), L.BF16) == 1
    st = L.stream()
    w = torch.randn(128, 256, 1, 1, generator=g) / 16
    pk = torch.zeros(lib.dvae_conv_packed_elems(C.byref(d), 0), device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_conv_pack(w.cuda().data_ptr(), pk.data_ptr(), C.byref(d), 0, L.BF16, st))
    x = torch.randn(B, hw, hw, 256, generator=g).cuda().bfloat16()
    bias = torch.randn(128, generator=g).cuda()
    emb = torch.randn(ncls, 128 * hw * hw, generator=g).cuda()
    table = torch.zeros(ncls * hw * hw * 128, device="cuda", dtype=torch.bfloat16)
    L.check(lib.dvae_emb_pack(emb.data_ptr(), table.data_ptr(), ncls, hw * hw, 128, 1, L.BF16, st))
    y = torch.randint(0, ncls, (B,), generator=g).cuda()
    z_ref = torch.empty(B, hw, hw, 128, device="cuda", dtype=torch.bfloat16)
    ep = L.Epilogue(None, bias.data_ptr(), None, 0, 0)
    L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep), z_ref.data_ptr(), C.byref(d), L.BF16, st))
    e_ref = torch.empty_like(z_ref)
    L.check(lib.dvae_emb_add_fwd(table.data_ptr(), y.data_ptr(), z_ref.data_ptr(), e_ref.data_ptr(), B, hw * hw, 128, 1, L.BF16, st))
    z, e = torch.empty_like(z_ref), torch.empty_like(z_ref)
    ep2 = L.Epilogue(None, bias.data_ptr(), None, 0, 0, None, 0, None, table.data_ptr(), y.data_ptr(), e.data_ptr())
    L.check(lib.dvae_conv_fwd(x.data_ptr(), pk.data_ptr(), C.byref(ep2), z.data_ptr(), C.byref(d), L.BF16, st))
    torch.cuda.synchronize()
    assert torch.equal(z, z_ref) and torch.equal(e, e_ref)
    # a conv that cannot do it must say so rather than drop the request
    d2 = L.ConvDesc(B, hw, hw, 128, 128, 3, 3, 1, 1, 0)
    assert lib.dvae_conv_supports_emb_out(C.byref(d2), L.BF16) == 0
