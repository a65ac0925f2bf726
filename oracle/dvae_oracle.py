"""CPU oracle for the D-VAE hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, with stock fp32 ``torch`` CPU ops in a functional style, the algorithm of the
reference's rate-constrained disentangling VAE:

* network:        /root/reference/configs/models/resnet.py:1960-2259 (``ResNetDVAE``) and
                  ``BasicBlock`` :7-31
* loss + step:    /root/reference/main_vae_disentangle.py:180-211 (``VAE_Trainer.train_batch``)
* purification:   /root/reference/main_vae_disentangle.py:45-95 (``Trainer.train`` epoch 0)
* optimiser cfg:  /root/reference/configs/cifar10/resnet18DVAE.yaml:16-28

It is never imported by the product package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker / timed CPU baseline.

Pinning: the reference ships no tests and no golden vectors.  The pins are the fixtures under
``tests/golden/`` which ``tests/golden/make_golden.py`` generated in the build container by importing
the reference's own ``configs/models/resnet.py`` by file path and driving its sub-methods; the
oracle is checked against those fixtures by ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------
@dataclass
class DVAEConfig:
    num_classes: int = 10
    stem_end_block: int = 7          # 7 or 9 are the only working D-VAE variants of the reference
    latent_dim: int = 256
    sample_wise: bool = True         # resnet.py:1974
    use_y: bool = False              # published scripts run with use_y=False (store_false flag)
    spatial_emb: bool = True
    image_size: int = 32

    @property
    def latent_hw(self) -> int:
        return self.image_size // 2


def block_names(cfg: DVAEConfig):
    """BasicBlocks on the stem, in execution order (resnet.py:2139-2150)."""
    names = ["layer1.0", "layer1.1", "layer2.0"]
    if cfg.stem_end_block == 9:
        names.append("layer2.1")
    elif cfg.stem_end_block != 7:
        raise ValueError("only stem_end_block 7 and 9 build a working D-VAE in the reference")
    return names


def param_shapes(cfg: DVAEConfig):
    """name -> shape for every parameter/buffer on the hot path, reference state_dict names."""
    s = {}
    s["conv1.weight"] = (64, 3, 3, 3)

    def bn(prefix, c):
        s[prefix + ".weight"] = (c,)
        s[prefix + ".bias"] = (c,)
        s[prefix + ".running_mean"] = (c,)
        s[prefix + ".running_var"] = (c,)

    bn("bn1", 64)
    for blk in block_names(cfg):
        cin = 64
        cout = 64 if blk.startswith("layer1") else 128
        if blk == "layer2.1":
            cin = 128
        s[blk + ".conv1.weight"] = (cout, cin, 3, 3)
        bn(blk + ".bn1", cout)
        s[blk + ".conv2.weight"] = (cout, cout, 3, 3)
        bn(blk + ".bn2", cout)
        if blk == "layer2.0":
            s[blk + ".shortcut.0.weight"] = (cout, cin, 1, 1)
            bn(blk + ".shortcut.1", cout)
    L = cfg.latent_dim
    s["fc_mu.weight"] = (L, 128, 1, 1)
    s["fc_mu.bias"] = (L,)
    s["fc_var.weight"] = (L, 128, 1, 1)
    s["fc_var.bias"] = (L,)
    s["post_conv.weight"] = (128, L, 1, 1)
    s["post_conv.bias"] = (128,)
    for d in ("decoder", "decoder_p"):
        s[d + "1.0.weight"] = (128, 64, 4, 4)
        s[d + "1.0.bias"] = (64,)
        s[d + "1.1.weight"] = (64, 64, 3, 3)
        s[d + "1.1.bias"] = (64,)
        s[d + "2.0.weight"] = (64, 32, 3, 3)
        s[d + "2.0.bias"] = (32,)
        s[d + "3.0.weight"] = (32, 3, 3, 3)
        s[d + "3.0.bias"] = (3,)
    hw = cfg.latent_hw
    s["emb_p.weight"] = (cfg.num_classes, 128 * hw * hw if cfg.spatial_emb else 128)
    return s


def is_buffer(name: str) -> bool:
    return name.endswith("running_mean") or name.endswith("running_var")


def recipe_params(cfg: DVAEConfig, scale_overrides: dict | None = None):
    """Deterministic closed-form parameter fill (no weight file).

    Every tensor is drawn from a numpy PCG64 stream seeded with crc32(name), so the oracle, the
    golden generator (which loads these into the reference module) and the HIP build all see the
    same numbers on every machine.  Magnitudes follow torch's default initialisers
    (kaiming_uniform(a=sqrt 5) => U(-1/sqrt(fan_in), 1/sqrt(fan_in)); Embedding ~ N(0,1)) while
    BatchNorm affine/running statistics are perturbed away from their trivial defaults.
    """
    return _recipe_fill(param_shapes(cfg), scale_overrides)


def _recipe_fill(shapes: dict, scale_overrides: dict | None = None):
    out = {}
    for name, shape in shapes.items():
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        if name.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            a = rng.uniform(-0.1, 0.1, shape)
        elif ".bn" in name or name.startswith("bn1") or ".shortcut.1" in name:
            a = rng.uniform(0.5, 1.5, shape) if name.endswith("weight") else rng.uniform(-0.2, 0.2, shape)
        elif name == "emb_p.weight":
            a = rng.standard_normal(shape)
        else:
            wname = name.rsplit(".", 1)[0] + ".weight"
            wshape = shapes[wname]
            fan_in = 1
            for v in wshape[1:]:
                fan_in *= v                               # torch's fan_in: size(1) * receptive field
            b = 1.0 / math.sqrt(fan_in)
            a = rng.uniform(-b, b, shape)
        if scale_overrides and name in scale_overrides:
            a = a * scale_overrides[name]
        out[name] = torch.tensor(a, dtype=torch.float32)
    return out


def make_inputs(cfg: DVAEConfig, batch: int, seed: int = 1234):
    """x ~ U[0,1) NCHW fp32, y = arange % classes, eps ~ N(0,1) with the latent's shape."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, cfg.image_size, cfg.image_size, generator=g)
    y = torch.arange(batch) % cfg.num_classes
    eps = torch.randn(batch, cfg.latent_dim, cfg.latent_hw, cfg.latent_hw, generator=g)
    return x, y, eps


# --------------------------------------------------------------------------------------------
# network (functional)
# --------------------------------------------------------------------------------------------
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
IN_EPS = 1e-5


class _MaskedReLU(torch.autograd.Function):
    """ReLU whose backward uses a caller-supplied mask instead of (output > 0).

    A ReLU's gradient is discontinuous at 0: when a pre-activation is within rounding of zero, two
    correct fp32 implementations may disagree on that one element's mask, and with small batches a
    single flipped element moves some weight gradients by ~1e-2.  Gradient parity is therefore
    checked with the masks of the implementation under test (each disagreement being verified to
    sit at an ambiguous pre-activation), so every other source of error is held to a tight bound."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


# --- bf16-storage emulation -------------------------------------------------------------------
# The device path's bf16 mode keeps activations, gradient maps and packed weights in bf16 and everything else (sums,
# statistics, parameters, parameter gradients, losses) in fp32.  Inside ``with bf16_storage():`` the functions below
# round to bf16 at exactly those storage points -- forward values and the gradients that flow back through them -- so that
# the bf16 path can be held to a reference that differs from it by summation order only, instead of to the fp32 result
# through a loose bound.  Outside the block nothing changes (the reference is fp32 throughout).
_BF16_STORAGE = [False]


class bf16_storage:
    def __enter__(self):
        self.prev, _BF16_STORAGE[0] = _BF16_STORAGE[0], True

    def __exit__(self, *a):
        _BF16_STORAGE[0] = self.prev


def _r16(t):
    return t.bfloat16().float()


class _RoundBoth(torch.autograd.Function):
    """A tensor stored in bf16: the value is rounded, and so is the gradient that arrives for it."""

    @staticmethod
    def forward(ctx, x):
        return _r16(x)

    @staticmethod
    def backward(ctx, g):
        return _r16(g)


class _RoundFwd(torch.autograd.Function):
    """A packed bf16 copy of an fp32 master tensor (weights, the embedding table): rounded value, fp32 gradient."""

    @staticmethod
    def forward(ctx, x):
        return _r16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """An fp32 tensor whose gradient map is stored in bf16 (decoder outputs, mu | log_var)."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return _r16(g)


def _st(x):
    return _RoundBoth.apply(x) if _BF16_STORAGE[0] else x


def _wq(w):
    return _RoundFwd.apply(w) if _BF16_STORAGE[0] else w


def _gq(x):
    return _RoundBwd.apply(x) if _BF16_STORAGE[0] else x


def _relu(x, name, masks, pre):
    if pre is not None:
        pre[name] = x.detach()
    if masks is not None and name in masks:
        return _MaskedReLU.apply(x, masks[name])
    return F.relu(x)


def _bn(p, prefix, x, train: bool, new_stats: dict | None):
    """nn.BatchNorm2d: batch statistics in train mode (biased var normalises, unbiased var goes
    into the running estimate with momentum 0.1), running statistics in eval mode."""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    if train:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
        if new_stats is not None:
            new_stats[prefix + ".running_mean"] = rm2
            new_stats[prefix + ".running_var"] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def _basic_block(p, blk, x, train, new_stats, masks=None, pre=None):
    """BasicBlock.forward, resnet.py:26-31."""
    stride = 2 if blk in ("layer2.0", "layer3.0", "layer4.0") else 1
    out = _st(F.conv2d(x, _wq(p[blk + ".conv1.weight"]), None, stride, 1))
    out = _st(_relu(_bn(p, blk + ".bn1", out, train, new_stats), blk + ".bn1.out", masks, pre))
    out = _bn(p, blk + ".bn2", _st(F.conv2d(out, _wq(p[blk + ".conv2.weight"]), None, 1, 1)), train, new_stats)
    if (blk + ".shortcut.0.weight") in p:
        sc = _st(F.conv2d(x, _wq(p[blk + ".shortcut.0.weight"]), None, stride, 0))
        sc = _st(_bn(p, blk + ".shortcut.1", sc, train, new_stats))
    else:
        sc = x
    return _st(_relu(out + sc, blk + ".out", masks, pre))


def forward_stem(p, cfg, x, train, new_stats=None, masks=None, pre=None):
    """resnet.py:2139-2150."""
    out = _st(F.conv2d(_st(x), _wq(p["conv1.weight"]), None, 1, 1))
    out = _st(_relu(_bn(p, "bn1", out, train, new_stats), "bn1.out", masks, pre))
    for blk in block_names(cfg):
        out = _basic_block(p, blk, out, train, new_stats, masks, pre)
    return out


def _decoder(p, d, z, sigmoid: bool, masks=None, pre=None):
    """resnet.py:2011-2028 (decoder) / :2033-2050 (decoder_p, no sigmoid)."""
    h = _st(F.conv_transpose2d(z, _wq(p[d + "1.0.weight"]), p[d + "1.0.bias"], stride=2, padding=1))
    h = _st(F.conv_transpose2d(h, _wq(p[d + "1.1.weight"]), p[d + "1.1.bias"], stride=1, padding=1))
    h = _st(_relu(F.instance_norm(h, eps=IN_EPS), d + ".d3", masks, pre))
    h = _st(F.conv_transpose2d(h, _wq(p[d + "2.0.weight"]), p[d + "2.0.bias"], stride=1, padding=1))
    h = _st(_relu(F.instance_norm(h, eps=IN_EPS), d + ".d5", masks, pre))
    h = _gq(F.conv_transpose2d(h, _wq(p[d + "3.0.weight"]), p[d + "3.0.bias"], stride=1, padding=1))
    return torch.sigmoid(h) if sigmoid else h


def forward(p, cfg: DVAEConfig, x, y, eps, train: bool, new_stats: dict | None = None, masks=None, pre=None):
    """ResNetDVAE.forward (resnet.py:2203-2222) minus the unused classifier head.

    ``eps`` replaces the ``torch.randn_like`` draw of ``reparameterize`` (:2257-2259); the additive
    ``randn * 0.0`` on the input (:2206-2207) is a numeric no-op and is dropped.
    """
    stem_out = forward_stem(p, cfg, x, train, new_stats, masks, pre)
    mu = _gq(F.conv2d(stem_out, _wq(p["fc_mu.weight"]), p["fc_mu.bias"]))
    log_var = _gq(F.conv2d(stem_out, _wq(p["fc_var.weight"]), p["fc_var.bias"]))
    z1 = _st(eps * torch.exp(0.5 * log_var) + mu)
    z = _st(F.conv2d(z1, _wq(p["post_conv.weight"]), p["post_conv.bias"]))
    rec = _decoder(p, "decoder", z, True, masks, pre)
    table = _wq(p["emb_p.weight"])
    if cfg.spatial_emb:
        emb = table[y].reshape(stem_out.shape)
    else:
        emb = table[y][:, :, None, None].repeat(1, 1, stem_out.shape[2], stem_out.shape[3])
    if cfg.sample_wise:
        dec_in = _st(emb + (stem_out if cfg.use_y else z))
    else:
        dec_in = _st(emb)
    perturbation = _decoder(p, "decoder_p", dec_in, False, masks, pre)
    return {"stem_out": stem_out, "mu": mu, "log_var": log_var, "z1": z1, "z": z,
            "rec": rec, "perturbation": perturbation}


# --------------------------------------------------------------------------------------------
# loss and optimiser step
# --------------------------------------------------------------------------------------------
def stage_constants(stage: int, kd: float = 3.0):
    """(kld_target, kld_weight) per main_vae_disentangle.py:181-184 and :200-203."""
    target = 1.0 if stage == 0 else kd
    weight = 0.01 if target > 0.99 else 0.5
    return target, weight


def losses(out, x, epoch: int, kld_target: float, kld_weight: float):
    """main_vae_disentangle.py:190-207.  Python's builtin ``max(tensor, float)`` keeps the tensor
    (and therefore its gradient) unless ``float > tensor`` holds strictly."""
    rec, p, mu, lv = out["rec"], out["perturbation"], out["mu"], out["log_var"]
    loss_rec = F.mse_loss(rec, x)
    loss_rec_p = F.mse_loss((rec + p).clamp(0, 1), x)
    loss_p = F.mse_loss(p, torch.zeros_like(p))
    kld = torch.mean(-0.5 * torch.sum(1 + lv - mu ** 2 - lv.exp(), dim=1), dim=(0, 1, 2))
    loss = loss_rec + loss_rec_p + kld_weight * max(kld, kld_target)
    if epoch > 10:
        loss = loss + max(loss_p, 0.002)
    return {"loss": loss, "loss_rec": loss_rec, "loss_rec_p": loss_rec_p, "loss_p": loss_p, "kld": kld}


def trainable_names(p):
    return [k for k in p if not is_buffer(k)]


def clip_grad_norm(grads: dict, max_norm: float):
    """torch.nn.utils.clip_grad_norm_: L2 norm of the per-tensor L2 norms, coef clamped to 1."""
    norms = torch.stack([g.norm(2) for g in grads.values()])
    total = norms.norm(2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def adam_step(p, grads, state, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (single-tensor form), weight_decay 0, amsgrad off."""
    b1, b2 = betas
    for k, g in grads.items():
        st = state.setdefault(k, {"step": 0, "m": torch.zeros_like(g), "v": torch.zeros_like(g)})
        st["step"] += 1
        st["m"] = st["m"] * b1 + (1 - b1) * g
        st["v"] = st["v"] * b2 + (1 - b2) * g * g
        bc1 = 1 - b1 ** st["step"]
        bc2 = 1 - b2 ** st["step"]
        denom = st["v"].sqrt() / math.sqrt(bc2) + eps
        p[k] = p[k] - (lr / bc1) * st["m"] / denom


def cosine_lr(base_lr: float, epoch: int, t_max: int, eta_min: float = 0.0):
    """Closed form of torch.optim.lr_scheduler.CosineAnnealingLR stepped once per epoch."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


def train_step(p, cfg, x, y, eps, adam_state, epoch, stage, kd=3.0, lr=5e-3, grad_clip=5.0, masks=None, pre=None):
    """One VAE_Trainer.train_batch without the model_c monitors.  Mutates ``p`` and ``adam_state``.
    Returns scalars and the clipped-before gradients for inspection."""
    target, weight = stage_constants(stage, kd)
    names = trainable_names(p)
    leaf = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    new_stats = {}
    out = forward(leaf, cfg, x, y, eps, True, new_stats, masks, pre)
    L = losses(out, x, epoch, target, weight)
    L["loss"].backward()
    grads = {k: leaf[k].grad.detach() for k in names if leaf[k].grad is not None}
    total, clipped = clip_grad_norm(grads, grad_clip)
    adam_step(p, clipped, adam_state, lr)
    for k, v in new_stats.items():
        p[k] = v.detach()
    info = {k: float(v.detach()) for k, v in L.items()}
    info["grad_norm"] = float(total)
    return info, grads, {k: v.detach() for k, v in out.items()}


def train_step_data_parallel(p, cfg, x, y, eps, adam_state, epoch, stage, world, kd=3.0, lr=5e-3, grad_clip=5.0,
                             masks=None):
    """One VAE_Trainer.train_batch as ``torch.nn.DataParallel`` over ``world`` replicas runs it
    (main_vae_disentangle.py:426-428): the batch is scattered in contiguous chunks, every replica normalises with the
    BatchNorm statistics of ITS chunk, the outputs are gathered and the loss block (:190-207) sees the whole batch;
    autograd sums the replicas' parameter gradients.  The module's BatchNorm buffers after the step are replica 0's
    (the replicas on the other devices update copies that are dropped).  ``masks``: optional list (one dict per
    replica) of ReLU masks, as in ``train_step``.  Mutates ``p`` and ``adam_state``; returns (scalars, gradients,
    per-replica new BatchNorm statistics)."""
    target, weight = stage_constants(stage, kd)
    names = trainable_names(p)
    leaf = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    b = x.shape[0]
    assert b % world == 0
    per = b // world
    outs, stats = [], []
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        st = {}
        outs.append(forward(leaf, cfg, x[sl], y[sl], eps[sl], True, st, masks[r] if masks is not None else None))
        stats.append({k: v.detach() for k, v in st.items()})
    out = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    L = losses(out, x, epoch, target, weight)
    L["loss"].backward()
    grads = {k: leaf[k].grad.detach() for k in names if leaf[k].grad is not None}
    total, clipped = clip_grad_norm(grads, grad_clip)
    adam_step(p, clipped, adam_state, lr)
    for k, v in stats[0].items():
        p[k] = v
    info = {k: float(v.detach()) for k, v in L.items()}
    info["grad_norm"] = float(total)
    return info, grads, stats


# --------------------------------------------------------------------------------------------
# purification
# --------------------------------------------------------------------------------------------
def quantize_u8(img_chw: torch.Tensor) -> np.ndarray:
    """np.clip(255 * x.permute(1,2,0).numpy(), 0, 255).astype(np.uint8)  (main_vae_disentangle.py:67):
    the fp32 product is truncated toward zero."""
    return np.clip(255 * img_chw.permute(1, 2, 0).numpy(), a_min=0, a_max=255).astype(np.uint8)


def to_tensor_u8(data_nhwc_u8: np.ndarray) -> torch.Tensor:
    """torchvision ToTensor on a uint8 HWC array: CHW float32 / 255."""
    return torch.from_numpy(data_nhwc_u8).permute(0, 3, 1, 2).to(torch.float32).div(255)


@torch.no_grad()
def purify(p, cfg, data_u8, labels, eps_fn, stage: int, batch: int = 128):
    """Trainer.train(epoch=0): sequential pass, eval-mode model (BN running stats, InstanceNorm
    per-sample, reparameterize still sampling).  ``eps_fn(pass_index, lo, hi)`` supplies the
    injected noise for images [lo, hi).  Returns (uint8 NHWC purified set, float64 perturbations)."""
    n = data_u8.shape[0]
    out_u8 = np.empty_like(data_u8)
    noise = np.zeros((n, 3, cfg.image_size, cfg.image_size), dtype=np.float64)
    for lo in range(0, n, batch):
        hi = min(n, lo + batch)
        x = to_tensor_u8(data_u8[lo:hi])
        yb = torch.as_tensor(labels[lo:hi], dtype=torch.long)
        o = forward(p, cfg, x, yb, eps_fn(0, lo, hi), False)
        pert = o["perturbation"]
        img = (x - pert).clamp(0, 1)
        if stage > 0:
            o = forward(p, cfg, img, yb, eps_fn(1, lo, hi), False)
            pert = o["perturbation"]
            img = o["rec"].clamp(0, 1)
        for k in range(hi - lo):
            out_u8[lo + k] = quantize_u8(img[k])
            noise[lo + k] = pert[k].numpy()
    return out_u8, noise


# --------------------------------------------------------------------------------------------
# unlearnable-example ingress
# --------------------------------------------------------------------------------------------
def poison_u8(data_u8: np.ndarray, perturbation: torch.Tensor, targets=None, poison_samples_idx=None) -> np.ndarray:
    """PoisonCIFAR10.__init__ (dataset.py:1235, :1241, :1287-1289) for full-image noise: the perturbation tensor is
    scaled by 255 and clamped to [-255, 255] in its own dtype (:1235) and permuted to HWC; patch_noise_extend_to_img
    writes the image's noise into a float32 mask (:534, :550; for full-image noise the mask is the noise, rounded to
    float32); each selected image becomes uint8(clip(float32(image) + mask, 0, 255)).  ``targets`` given =
    class-wise (row = label), else sample-wise."""
    pt = perturbation.clone().mul(255).clamp_(-255, 255).permute(0, 2, 3, 1).numpy()
    out = data_u8.astype(np.float32)
    idx = range(len(out)) if poison_samples_idx is None else poison_samples_idx
    for i in idx:
        noise = np.zeros(out[i].shape, np.float32)
        noise[...] = pt[int(targets[i])] if targets is not None else pt[i]
        out[i] = out[i] + noise
        out[i] = np.clip(out[i], a_min=0, a_max=255)
    return out.astype(np.uint8)


# --------------------------------------------------------------------------------------------
# classifier stage: CIFAR ResNet-18 (resnet.py:62-122) and its train step (main_vae_disentangle.py:97-125)
# --------------------------------------------------------------------------------------------
RESNET18_BLOCKS = [f"layer{i}.{j}" for i in (1, 2, 3, 4) for j in (0, 1)]


def resnet18_shapes(num_classes: int = 10):
    """name -> shape of ResNet18's parameters and BatchNorm statistics, reference state_dict names."""
    s = {"conv1.weight": (64, 3, 3, 3)}

    def bn(prefix, c):
        for k, shape in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,))):
            s[prefix + "." + k] = shape

    bn("bn1", 64)
    cin = 64
    for blk in RESNET18_BLOCKS:
        c = 64 << (int(blk[5]) - 1)
        s[blk + ".conv1.weight"] = (c, cin, 3, 3)
        bn(blk + ".bn1", c)
        s[blk + ".conv2.weight"] = (c, c, 3, 3)
        bn(blk + ".bn2", c)
        if cin != c:
            s[blk + ".shortcut.0.weight"] = (c, cin, 1, 1)
            bn(blk + ".shortcut.1", c)
        cin = c
    s["linear.weight"] = (num_classes, 512)
    s["linear.bias"] = (num_classes,)
    return s


def resnet18_recipe(num_classes: int = 10, scale_overrides: dict | None = None):
    """The closed-form fill of recipe_params for the classifier's tensors."""
    return _recipe_fill(resnet18_shapes(num_classes), scale_overrides)


def resnet18_forward(p, x, train: bool, new_stats: dict | None = None, masks=None, pre=None):
    """ResNet.forward (resnet.py:107-118): conv1-bn1-relu, four stages of two BasicBlocks, 4x4 average pool, linear."""
    out = _st(F.conv2d(_st(x), _wq(p["conv1.weight"]), None, 1, 1))
    out = _st(_relu(_bn(p, "bn1", out, train, new_stats), "bn1.out", masks, pre))
    for blk in RESNET18_BLOCKS:
        out = _basic_block(p, blk, out, train, new_stats, masks, pre)
    feat = out
    out = F.adaptive_avg_pool2d(_gq(out), 1).flatten(1)   # (the trunk's output leaves the device path as fp32)
    return F.linear(out, p["linear.weight"], p["linear.bias"]), feat


def resnet18_train_grads(p, x, y, masks=None):
    """Forward in train mode, cross-entropy, backward (main_vae_disentangle.py:100-107): logits, loss, gradients by
    name and the BatchNorm running statistics after the step."""
    p = {k: (v.clone().requires_grad_(True) if not is_buffer(k) else v.clone()) for k, v in p.items()}
    stats = {}
    logits, _ = resnet18_forward(p, x, True, stats, masks)
    loss = F.cross_entropy(logits, y)
    names = [k for k in p if not is_buffer(k)]
    grads = torch.autograd.grad(loss, [p[k] for k in names])
    return logits.detach(), float(loss.detach()), dict(zip(names, grads)), stats
