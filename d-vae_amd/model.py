"""Drop-in ``ResNet18DVAE`` for the reference's D-VAE model API.

Mirrors the interface of ``ResNetDVAE`` in /root/reference/configs/models/resnet.py:1960-2262:
same constructor, flags (``sample_wise``, ``use_y``, ``spatial_emb``), ``set_spatial_emb()``,
``forward(x, y) -> (logits, rec, mu, log_var, perturbation)``, ``purify_with_mean``, ``encode``,
``reparameterize``, ``forward_stem``, ``forward_main_branch``, ``forward_ae_branch``, and a
``state_dict()`` with the reference's 161 keys (including the aliased ``decoder.*`` /
``decoder_p.*`` entries), so ``model_s{it}.pth`` files are interchangeable.

The module tree only *holds* parameters.  Everything on the hot path executes in libdvae_hip through
``DVAEEngine``; the parameters on the path are views into the engine's flat fp32 bucket.  There is no
CPU or stock-torch fallback for the path: calling ``forward`` without a GPU / the built library raises.
The classifier head of ``model_s`` (``forward_main_branch``) never enters the D-VAE loss; it is kept
as stock torch modules and is evaluated only when ``compute_logits`` is set.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .engine import DVAEConfig, DVAEEngine, bn_names, param_layout


class _ResBlock(nn.Module):
    """Parameter container laid out like the reference's BasicBlock (conv1/bn1/conv2/bn2/shortcut);
    its stock-torch forward serves the off-path classifier head only."""
    expansion = 1

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != cout:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y)) + self.shortcut(x)
        return F.relu(y)


def _decoder_stack():
    d1 = nn.Sequential(nn.ConvTranspose2d(128, 64, 4, stride=2, padding=1), nn.ConvTranspose2d(64, 64, 3, stride=1, padding=1),
                       nn.InstanceNorm2d(64), nn.ReLU())
    d2 = nn.Sequential(nn.ConvTranspose2d(64, 32, 3, stride=1, padding=1), nn.InstanceNorm2d(32), nn.ReLU())
    return d1, d2


class _PathFunction(torch.autograd.Function):
    """The whole hot path as one autograd node: forward runs the engine, backward receives the
    gradients of (rec, mu, log_var, perturbation) and leaves parameter gradients in the bucket."""

    @staticmethod
    def forward(ctx, model, x, y, eps, anchor):
        eng = model._engine_for(x)
        b = x.shape[0]
        x0 = eng.input_from_nchw(x.detach().float())
        e = eng.eps_from_nchw(eps) if eps is not None else eng.eps_philox(b, first=model._draws * b, stream_id=7)
        model._draws += 1
        train = model.training
        # mu / log_var are returned to the caller: they must be written (the fused inference head keeps them in registers)
        rec_pre, p = eng.forward(x0, y, e, b, train, need_mulv=True)
        S, h, Ld = eng.S, eng.h, eng.Ld
        rec_act = eng.buf("rec", (eng.B, S, S, 4), torch.float32)
        L.check(eng.lib.dvae_recon_loss_fwd(L.ptr(rec_pre), L.ptr(p), L.ptr(x0), L.ptr(rec_act), L.ptr(eng.recon_partial),
                                            b * S * S, eng.dtype, L.stream()))
        mulv = eng.act("mulv", h, 2 * Ld, f32=True)
        if eps is None and not train and eng.dtype == L.BF16:
            mulv = eng.act("mulv.act", h, 2 * Ld)       # inference keeps mu / log_var in the act dtype
        rec = eng.to_nchw(rec_act, b, 3, S)
        pert = eng.to_nchw(p, b, 3, S)
        mu = eng.to_nchw(mulv, b, Ld, h, 0)
        lv = eng.to_nchw(mulv, b, Ld, h, Ld)
        if train:
            model._nbt += 1
        ctx.model, ctx.eng, ctx.b = model, eng, b
        ctx.save_for_backward(rec)
        return rec, mu, lv, pert

    @staticmethod
    def backward(ctx, d_rec, d_mu, d_lv, d_pert):
        eng, b, model = ctx.eng, ctx.b, ctx.model
        (rec,) = ctx.saved_tensors
        S, h, Ld = eng.S, eng.h, eng.Ld
        zero = lambda t, shape: torch.zeros(shape, device=eng.dev) if t is None else t.contiguous().float()   # noqa: E731
        d_rec = zero(d_rec, rec.shape)
        d_pert = zero(d_pert, rec.shape)
        d_rec_pre = eng.act("g.decoder.d6", S, 3)
        L.check(eng.lib.dvae_nchw_to_act(L.ptr((d_rec * rec * (1 - rec)).contiguous()), L.ptr(d_rec_pre), b, 3, S, S, eng.dtype, L.stream()))
        d_p = eng.act("g.decoder_p.d6", S, 3)
        L.check(eng.lib.dvae_nchw_to_act(L.ptr(d_pert), L.ptr(d_p), b, 3, S, S, eng.dtype, L.stream()))
        extra = None
        if d_mu is not None or d_lv is not None:
            dm = zero(d_mu, (b, Ld, h, h))
            dl = zero(d_lv, (b, Ld, h, h))
            extra = torch.zeros_like(eng.act("g.mulv", h, 2 * Ld))
            extra[:b] = torch.cat([dm, dl], 1).permute(0, 2, 3, 1).to(extra.dtype)
        eng.scal.zero_()            # no fused-loss KL term on this path: d_mu / d_lv arrive from autograd
        attached = model.conv1.weight.grad is not None     # optimizer.zero_grad(set_to_none=True) detaches
        saved = eng.G.clone() if (model._accumulate and attached) else None
        eng.backward(d_rec_pre, d_p, extra)
        if saved is not None:
            eng.G.add_(saved)
        if not attached:
            named_p = dict(model.named_parameters())
            for name, _ in param_layout(eng.cfg):
                named_p[name].grad = eng.grad[name]
        model._accumulate = True
        return None, None, None, None, torch.zeros_like(model._anchor)


def _dtype_code(d):
    """"f32" / "bf16" (or the library's DVAE_F32 / DVAE_BF16 codes) -> library code."""
    if d in (L.F32, L.BF16) and not isinstance(d, str):
        return int(d)
    try:
        return {"f32": L.F32, "fp32": L.F32, "float32": L.F32, "bf16": L.BF16, "bfloat16": L.BF16}[str(d).lower()]
    except KeyError:
        raise ValueError(f"act_dtype must be 'f32' or 'bf16', got {d!r}") from None


class ResNetDVAE(nn.Module):
    """``act_dtype`` (not in the reference, which is fp32 throughout) selects the arithmetic of the path: "f32" = exact
    fp32 MFMA, "bf16" = bf16 activations / packed weights with fp32 accumulation, statistics, losses, master
    parameters and optimiser state.  It can also be set as an attribute at any time before the next call; the
    environment variable DVAE_ACT_DTYPE overrides the constructor default."""

    def __init__(self, num_blocks=(2, 2, 2, 2), num_classes=10, stem_end_block=13, latent_dim=128, act_dtype=None):
        super().__init__()
        if stem_end_block not in (7, 9):
            raise ValueError("stem_end_block must be 7 or 9: the reference's other values do not build a working D-VAE "
                             "(no decoder_p for 11/13, channel mismatches for 5/15/17)")
        self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (c, s) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2))):
            blocks = []
            for j in range(num_blocks[i]):
                blocks.append(_ResBlock(cin, c, s if j == 0 else 1))
                cin = c
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.linear = nn.Linear(512, num_classes)
        self.pooling = nn.AdaptiveAvgPool2d((1, 1))
        self.num_blocks = list(num_blocks)
        self.latent_dim = latent_dim
        self.sample_wise = True
        self.use_y = True
        self.spatial_emb = False
        self.num_classes = num_classes
        self.stem_end_block = stem_end_block
        self.encoder = nn.Identity()
        self.fc_mu = nn.Conv2d(128, latent_dim, 1)
        self.fc_var = nn.Conv2d(128, latent_dim, 1)
        self.post_conv = nn.Conv2d(latent_dim, 128, 1)
        self.decoder1, self.decoder2 = _decoder_stack()
        self.decoder3 = nn.Sequential(nn.ConvTranspose2d(32, 3, 3, stride=1, padding=1), nn.Sigmoid())
        self.decoder = nn.Sequential(self.decoder1, self.decoder2, self.decoder3)
        self.emb_p = nn.Embedding(num_classes, 128)
        self.decoder_p1, self.decoder_p2 = _decoder_stack()
        self.decoder_p3 = nn.Sequential(nn.ConvTranspose2d(32, 3, 3, stride=1, padding=1))
        self.decoder_p = nn.Sequential(self.decoder_p1, self.decoder_p2, self.decoder_p3)
        # engine state (not part of state_dict)
        self.compute_logits = False
        self.act_dtype = _dtype_code(act_dtype if act_dtype is not None else os.environ.get("DVAE_ACT_DTYPE", "f32"))
        self.image_size = 32
        self._engine = None
        self._versions = None
        self._draws = 0
        self._accumulate = False
        # a plain (unregistered) leaf that makes autograd record the path node even though the path's
        # parameters are not inputs of the autograd.Function
        object.__setattr__(self, "_anchor", torch.zeros((), requires_grad=True))

    # ---- reference API --------------------------------------------------------------------------
    def set_spatial_emb(self):
        """resnet.py:2125-2129: re-create emb_p after the flags were set."""
        width = 128 * 16 * 16 if self.spatial_emb else 128
        dev = self.emb_p.weight.device
        self.emb_p = nn.Embedding(self.num_classes, width).to(dev)
        self._engine = None

    def engine_config(self):
        return DVAEConfig(self.num_classes, self.stem_end_block, self.latent_dim, bool(self.sample_wise), bool(self.use_y),
                          bool(self.spatial_emb), self.image_size)

    def _named_path_tensors(self):
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        return sd

    def _engine_for(self, x):
        """The engine bound to this module's parameters (built on first use; rebuilt if the module was
        moved, resized or re-flagged).  Path parameters become views of the engine's flat bucket."""
        if not x.is_cuda:
            raise RuntimeError("the D-VAE hot path runs on libdvae_hip (MI355X) only; move the model and inputs to the GPU")
        b, S = x.shape[0], x.shape[-1]
        eng = self._engine
        dtype = _dtype_code(self.act_dtype)
        ok = (eng is not None and eng.B >= b and eng.S == S and eng.dev == x.device and eng.cfg == self._cfg_for(S)
              and eng.dtype == dtype and self.conv1.weight.data_ptr() == eng.param["conv1.weight"].data_ptr())
        if ok:
            # Parameters / BatchNorm buffers written behind the engine's back (load_state_dict, a stock optimiser's
            # step, an in-place edit) bump their tensors' version counters: the packed weights and the folded
            # BatchNorm are re-made when any of them moved since the last call.
            v = self._path_versions()
            if v != self._versions:
                eng.invalidate()
                self._versions = v
            return eng
        self.image_size = S
        cfg = self._cfg_for(S)
        old = {k: v.detach().clone() for k, v in self._named_path_tensors().items()}
        prev = eng
        eng = DVAEEngine(cfg, max(b, prev.B if prev is not None else 0), x.device, dtype,
                         seed=prev.seed if prev is not None else 0)
        eng.load_params(old, strict=True)
        if prev is not None and prev.nparam == eng.nparam and prev.dev == eng.dev and prev.cfg == cfg:
            # a resize or a dtype change in the middle of a run keeps the optimiser going: Adam moments, the step
            # count and the hyper-parameters live in the engine
            eng.M.copy_(prev.M)
            eng.V.copy_(prev.V)
            eng.hyper.copy_(prev.hyper)
            eng._hyper_host.copy_(prev._hyper_host)
        named_p = dict(self.named_parameters())
        for name, _ in param_layout(cfg):
            prm = named_p[name]
            prm.data = eng.param[name]
            prm.grad = eng.grad[name]
        mods = dict(self.named_modules())
        nb = bn_names(cfg)
        self._nbt = torch.zeros(len(nb), dtype=torch.long, device=x.device)
        for i, (name, _) in enumerate(nb):
            m = mods[name]
            self._nbt[i] = m.num_batches_tracked.to(x.device)
            m._buffers["running_mean"] = eng.buffers[name + ".running_mean"]
            m._buffers["running_var"] = eng.buffers[name + ".running_var"]
            m._buffers["num_batches_tracked"] = self._nbt[i]
        self._engine = eng
        self._accumulate = False
        self._versions = self._path_versions()
        return eng

    def _path_versions(self):
        named = self._named_path_tensors()
        return tuple(named[n]._version for n in self._path_names())

    def _path_names(self):
        names = getattr(self, "_path_names_cache", None)
        if names is None or names[0] != self._cfg_for(self.image_size):
            cfg = self._cfg_for(self.image_size)
            ns = [n for n, _ in param_layout(cfg)]
            for n, _ in bn_names(cfg):
                ns += [n + ".running_mean", n + ".running_var"]
            names = self._path_names_cache = (cfg, ns)
        return names[1]

    def _cfg_for(self, S):
        c = self.engine_config()
        c.image_size = S
        return c

    def zero_grad(self, set_to_none: bool = False):
        """Gradients of the path live in the engine's bucket; zeroing means the next backward overwrites."""
        self._accumulate = False
        if self._engine is not None:
            self._engine.G.zero_()
        for n, p in self.named_parameters():
            if p.grad is not None and (self._engine is None or n not in self._engine.grad):
                p.grad = None if set_to_none else p.grad.zero_()

    def forward(self, x, y, eps=None):
        """(logits, rec, mu, log_var, perturbation), resnet.py:2203-2222.  ``eps`` optionally injects the
        reparameterisation noise (NCHW, latent shape); by default it is drawn on the device (Philox)."""
        rec, mu, lv, pert = _PathFunction.apply(self, x, y, eps, self._anchor)
        if self.compute_logits:
            logits = self.forward_main_branch(self.forward_stem(x))
        else:
            logits = torch.zeros(x.shape[0], self.num_classes, device=x.device)
        return logits, rec, mu, lv, pert

    def purify_with_mean(self, x):
        """resnet.py:2224-2235: z = mu, i.e. the reparameterisation noise is zero."""
        b = x.shape[0]
        h = x.shape[-1] // 2
        y = torch.zeros(b, dtype=torch.long, device=x.device)
        zero = torch.zeros(b, self.latent_dim, h, h, device=x.device)
        rec, mu, lv, _ = _PathFunction.apply(self, x, y, zero, self._anchor)
        logits = self.forward_main_branch(self.forward_stem(x)) if self.compute_logits else torch.zeros(b, self.num_classes, device=x.device)
        return logits, rec, mu, lv

    @torch.no_grad()
    def forward_stem(self, x):
        eng = self._engine_for(x)
        b = x.shape[0]
        out = eng.stem(eng.input_from_nchw(x.float()), b, self.training)
        return eng.to_nchw(out, b, 128, eng.h)

    @torch.no_grad()
    def encode(self, input):
        eng = self._engine_for(input)
        b, h = input.shape[0], input.shape[-1]
        if h != eng.h:
            raise ValueError("encode expects the stem output of the configured image size")
        a = eng.act("layer2.0.out" if self.stem_end_block == 7 else "layer2.1.out", h, 128)
        L.check(eng.lib.dvae_nchw_to_act(L.ptr(input.contiguous().float()), L.ptr(a), b, 128, h, h, eng.dtype, L.stream()))
        if not eng.packed_valid:
            eng.pack_weights()
        mulv = eng.act("mulv", h, 2 * eng.Ld, f32=True)
        eng._conv("fc", a, mulv, b, shift=eng.param["fc.bias"], out_f32=True)
        return [eng.to_nchw(mulv, b, eng.Ld, h, 0), eng.to_nchw(mulv, b, eng.Ld, h, eng.Ld)]

    def reparameterize(self, mu, logvar):
        """resnet.py:2249-2259 on caller-held tensors (stock elementwise ops; the fused path does this in-kernel)."""
        return torch.randn_like(mu) * torch.exp(0.5 * logvar) + mu

    @torch.no_grad()
    def forward_ae_branch(self, stem_out, eps=None):
        eng = self._engine_for(stem_out)
        b, h = stem_out.shape[0], stem_out.shape[-1]
        a = eng.act("layer2.0.out" if self.stem_end_block == 7 else "layer2.1.out", h, 128)
        L.check(eng.lib.dvae_nchw_to_act(L.ptr(stem_out.contiguous().float()), L.ptr(a), b, 128, h, h, eng.dtype, L.stream()))
        e = eng.eps_from_nchw(eps) if eps is not None else eng.eps_philox(b, first=self._draws * b, stream_id=7)
        self._draws += 1
        mulv, _, z = eng.latent(a, e, b, keep_eps=False)
        rec_pre = eng._decoder("decoder", z, b, self.training)
        rec = torch.sigmoid(eng.to_nchw(rec_pre, b, 3, eng.S))
        return rec, eng.to_nchw(mulv, b, eng.Ld, h, 0), eng.to_nchw(mulv, b, eng.Ld, h, eng.Ld), eng.to_nchw(z, b, 128, h)

    def forward_main_branch(self, stem_out):
        """Classifier head (resnet.py:2169-2194), stock torch ops: off the D-VAE loss path."""
        x = stem_out
        if self.stem_end_block == 7:
            x = self.layer2[1](x)
        x = self.layer4(self.layer3(x))
        return self.linear(self.pooling(x).flatten(1))


def ResNet18DVAE(num_classes=10, stem_end_block=15, latent_dim=128, act_dtype=None):
    """Factory with the reference's signature (resnet.py:2261-2262) plus ``act_dtype`` ("f32" | "bf16")."""
    return ResNetDVAE((2, 2, 2, 2), num_classes=num_classes, stem_end_block=stem_end_block, latent_dim=latent_dim,
                      act_dtype=act_dtype)
