"""Static-schedule executor of the D-VAE hot path on one MI355X.

``DVAEEngine`` owns one flat fp32 bucket of the trainable parameters on the path (plus matching
gradient / Adam buffers), the packed GEMM images of every conv weight, and a dictionary of
preallocated NHWC activation buffers.  ``forward`` / ``backward`` / ``adam_step`` /
``purify_batch`` issue a fixed sequence of libdvae_hip launches on the current torch stream with no
allocation and no host synchronisation, so a whole train step can be captured in a HIP graph.

What it computes, by reference line:
  forward      configs/models/resnet.py:2139-2150, :2196-2201, :2208-2222 (classifier head elided:
               its logits never enter the loss, main_vae_disentangle.py:189-207)
  losses       main_vae_disentangle.py:190-192, :204-207
  backward     autograd of the above (main_vae_disentangle.py:209)
  adam_step    main_vae_disentangle.py:210-211, configs/cifar10/resnet18DVAE.yaml:16-28
  purify_batch main_vae_disentangle.py:59-69
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import _lib as L
from ._lib import BF16, F32, ConvDesc, Epilogue, check, ptr

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
IN_EPS = 1e-5


@dataclass
class DVAEConfig:
    num_classes: int = 10
    stem_end_block: int = 7
    latent_dim: int = 256
    sample_wise: bool = True
    use_y: bool = False
    spatial_emb: bool = True
    image_size: int = 32


@dataclass
class ConvSpec:
    name: str
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    transposed: bool
    hin: int          # spatial size of the module's input
    bias: bool

    @property
    def hout(self):
        if self.transposed:
            return (self.hin - 1) * self.stride - 2 * self.pad + self.k
        return (self.hin + 2 * self.pad - self.k) // self.stride + 1

    @property
    def wshape(self):
        return (self.cin, self.cout, self.k, self.k) if self.transposed else (self.cout, self.cin, self.k, self.k)


def stem_blocks(cfg):
    blocks = ["layer1.0", "layer1.1", "layer2.0"]
    if cfg.stem_end_block == 9:
        blocks.append("layer2.1")
    elif cfg.stem_end_block != 7:
        raise ValueError("stem_end_block must be 7 or 9 (the reference's other values do not build a working D-VAE)")
    return blocks


def conv_specs(cfg):
    """Every convolution on the path, keyed by the reference's module name."""
    S = cfg.image_size
    sp = {"conv1": ConvSpec("conv1", 3, 64, 3, 1, 1, False, S, False)}
    for blk in stem_blocks(cfg):
        if blk.startswith("layer1"):
            cin, cout, stride, hin = 64, 64, 1, S
        elif blk == "layer2.0":
            cin, cout, stride, hin = 64, 128, 2, S
        else:
            cin, cout, stride, hin = 128, 128, 1, S // 2
        sp[blk + ".conv1"] = ConvSpec(blk + ".conv1", cin, cout, 3, stride, 1, False, hin, False)
        sp[blk + ".conv2"] = ConvSpec(blk + ".conv2", cout, cout, 3, 1, 1, False, hin // stride, False)
        if stride != 1 or cin != cout:
            sp[blk + ".shortcut.0"] = ConvSpec(blk + ".shortcut.0", cin, cout, 1, stride, 0, False, hin, False)
    h = S // 2
    # fc_mu and fc_var read the same input: one 1x1 conv with 2L output channels
    sp["fc"] = ConvSpec("fc", 128, 2 * cfg.latent_dim, 1, 1, 0, False, h, True)
    sp["post_conv"] = ConvSpec("post_conv", cfg.latent_dim, 128, 1, 1, 0, False, h, True)
    for d in ("decoder", "decoder_p"):
        sp[d + "1.0"] = ConvSpec(d + "1.0", 128, 64, 4, 2, 1, True, h, True)
        sp[d + "1.1"] = ConvSpec(d + "1.1", 64, 64, 3, 1, 1, True, S, True)
        sp[d + "2.0"] = ConvSpec(d + "2.0", 64, 32, 3, 1, 1, True, S, True)
        sp[d + "3.0"] = ConvSpec(d + "3.0", 32, 3, 3, 1, 1, True, S, True)
    return sp


def bn_names(cfg):
    names = [("bn1", 64)]
    for blk in stem_blocks(cfg):
        c = 64 if blk.startswith("layer1") else 128
        names += [(blk + ".bn1", c), (blk + ".bn2", c)]
        if blk == "layer2.0":
            names.append((blk + ".shortcut.1", c))
    return names


def param_layout(cfg):
    """(name, shape) of every trainable tensor on the path, in flat-bucket order.  fc_mu/fc_var
    weights (and biases) are adjacent so that they form one [2L,128,1,1] conv weight."""
    sp = conv_specs(cfg)
    out = [("conv1.weight", sp["conv1"].wshape)]
    out += [("bn1.weight", (64,)), ("bn1.bias", (64,))]
    for blk in stem_blocks(cfg):
        c = sp[blk + ".conv1"].cout
        out += [(blk + ".conv1.weight", sp[blk + ".conv1"].wshape), (blk + ".bn1.weight", (c,)), (blk + ".bn1.bias", (c,)),
                (blk + ".conv2.weight", sp[blk + ".conv2"].wshape), (blk + ".bn2.weight", (c,)), (blk + ".bn2.bias", (c,))]
        if (blk + ".shortcut.0") in sp:
            out += [(blk + ".shortcut.0.weight", sp[blk + ".shortcut.0"].wshape),
                    (blk + ".shortcut.1.weight", (c,)), (blk + ".shortcut.1.bias", (c,))]
    Ld = cfg.latent_dim
    out += [("fc_mu.weight", (Ld, 128, 1, 1)), ("fc_var.weight", (Ld, 128, 1, 1)), ("fc_mu.bias", (Ld,)), ("fc_var.bias", (Ld,))]
    out += [("post_conv.weight", (128, Ld, 1, 1)), ("post_conv.bias", (128,))]
    for d in ("decoder", "decoder_p"):
        for l in ("1.0", "1.1", "2.0", "3.0"):
            out += [(d + l + ".weight", sp[d + l].wshape), (d + l + ".bias", (sp[d + l].cout,))]
    hw = cfg.image_size // 2
    out.append(("emb_p.weight", (cfg.num_classes, 128 * hw * hw if cfg.spatial_emb else 128)))
    return out


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


@dataclass
class PhiloxEps:
    """Reparameterisation noise to be drawn inside the reparameterise kernel: Philox4x32-10 keyed by
    (engine seed, stream_id) and counted by (global image index, element), so the draw for an image does
    not depend on batch composition, shard or world size."""
    first: int = 0
    index: torch.Tensor | None = None     # int32 [b] global image indices, or None: first + i
    stream_id: int = 0


class DVAEEngine:
    def __init__(self, cfg: DVAEConfig, max_batch: int, device="cuda", dtype: int = F32, seed: int = 0):
        if cfg.spatial_emb and cfg.image_size != 32:
            raise ValueError("spatial_emb fixes the embedding to 16x16 latents (32x32 inputs), as in the reference")
        self.cfg, self.B, self.dtype, self.seed = cfg, int(max_batch), dtype, int(seed)
        self.dev = torch.device(device)
        self.lib = L.lib()
        self.tdt = torch.float32 if dtype == F32 else torch.bfloat16
        self.specs = conv_specs(cfg)
        self.blocks = stem_blocks(cfg)
        self.S = cfg.image_size
        self.h = cfg.image_size // 2
        self.Ld = cfg.latent_dim
        # ---- flat parameter bucket -------------------------------------------------------------
        self.layout = []
        off = 0
        for name, shape in param_layout(cfg):
            n = _numel(shape)
            self.layout.append((name, shape, off, n))
            off += (n + 3) // 4 * 4
        self.nparam = off
        f32 = dict(device=self.dev, dtype=torch.float32)
        self.P = torch.zeros(off, **f32)
        self.G = torch.zeros(off, **f32)
        self.M = torch.zeros(off, **f32)
        self.V = torch.zeros(off, **f32)
        self.param = {n: self.P[o:o + k].view(s) for n, s, o, k in self.layout}
        self.grad = {n: self.G[o:o + k].view(s) for n, s, o, k in self.layout}
        # merged views for the fc_mu|fc_var conv
        o_w = dict((n, o) for n, _, o, _ in self.layout)
        k = self.Ld * 128
        self.param["fc.weight"] = self.P[o_w["fc_mu.weight"]:o_w["fc_mu.weight"] + 2 * k].view(2 * self.Ld, 128, 1, 1)
        self.grad["fc.weight"] = self.G[o_w["fc_mu.weight"]:o_w["fc_mu.weight"] + 2 * k].view(2 * self.Ld, 128, 1, 1)
        self.param["fc.bias"] = self.P[o_w["fc_mu.bias"]:o_w["fc_mu.bias"] + 2 * self.Ld]
        self.grad["fc.bias"] = self.G[o_w["fc_mu.bias"]:o_w["fc_mu.bias"] + 2 * self.Ld]
        # ---- BatchNorm buffers -----------------------------------------------------------------
        self.bns = bn_names(cfg)
        tot = sum(c for _, c in self.bns)
        self.R = torch.zeros(2 * tot, **f32)
        self.buffers = {}
        self.bn_fold = {}
        self.FOLD = torch.zeros(2 * tot, **f32)
        o = 0
        for n, c in self.bns:
            self.buffers[n + ".running_mean"] = self.R[o:o + c]
            self.buffers[n + ".running_var"] = self.R[tot + o:tot + o + c]
            self.bn_fold[n] = (self.FOLD[o:o + c], self.FOLD[tot + o:tot + o + c])
            o += c
        self.R[tot:].fill_(1.0)
        self.bn_save = {n: (torch.zeros(c, **f32), torch.zeros(c, **f32)) for n, c in self.bns}
        # ---- packed weights --------------------------------------------------------------------
        self.packed = {}
        for name, sp in self.specs.items():
            d = self._desc(sp, 1)
            for op in (L.OP_FWD, L.OP_DGRAD):
                if name == "conv1" and op == L.OP_DGRAD:
                    continue
                n = self.lib.dvae_conv_packed_elems(C.byref(d), op)
                self.packed[(name, op)] = torch.zeros(n, device=self.dev, dtype=self.tdt)
        self.emb_table = torch.zeros(self.param["emb_p.weight"].numel(), device=self.dev, dtype=self.tdt)
        # ---- scalars / partials ----------------------------------------------------------------
        self.hyper = torch.zeros(8, **f32)
        self._hyper_host = torch.zeros(8, dtype=torch.float32).pin_memory() if self.dev.type == "cuda" else torch.zeros(8)
        self.set_hyper(lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=5.0, grad_scale=1.0, steps_done=0)
        self.scal = torch.zeros(L.SCAL_COUNT, **f32)
        self.sums = torch.zeros(4, **f32)
        self.recon_partial = torch.zeros(3 * L.NPARTIAL, **f32)
        self.kl_partial = torch.zeros(L.NPARTIAL, **f32)
        self.norm_partial = torch.zeros(L.NPARTIAL, **f32)
        # ---- workspaces ------------------------------------------------------------------------
        self._bufs = {}
        self._descs = {}
        self._alloc_workspaces()
        self.fold_valid = False
        self.packed_valid = False
        self._fuse_in = all(self.lib.dvae_conv_supports_input_norm(C.byref(self._desc(self.specs[n], 1)), dtype)
                            for n in ("decoder2.0", "decoder3.0"))
        # the convs in front of the decoders' InstanceNorms can leave the statistics of their output behind
        self._fuse_stats = all(self.lib.dvae_conv_out_stats_floats(C.byref(self._desc(self.specs[n], 1)), dtype) > 0
                               for n in ("decoder1.1", "decoder2.0"))
        self._fuse_head = bool(self.lib.dvae_fc_reparam_supported(C.byref(self._desc(self.specs["fc"], 1)), self.Ld, dtype))
        # ... and the whole latent path (fc, reparameterisation, post_conv) can be one launch that never writes z
        self._fuse_head_post = bool(self._fuse_head and self.lib.dvae_fc_reparam_post_supported(
            C.byref(self._desc(self.specs["fc"], 1)), C.byref(self._desc(self.specs["post_conv"], 1)), self.Ld, dtype))
        # post_conv can leave decoder_p's input (z + class embedding) behind as a second output
        # (either table: the spatial one, or the plain class embedding -- one row for every pixel of an image)
        self._fuse_emb = bool(cfg.sample_wise and not cfg.use_y and
                              self.lib.dvae_conv_supports_emb_out(C.byref(self._desc(self.specs["post_conv"], 1)), dtype))
        # Purification draws its reparameterisation noise in-kernel.  bf16 engines take it from a 16-bit-uniform stream
        # (stream id bit 31: eight normals per Philox call, half the cost per normal; eps feeds a product that is rounded
        # to bf16 there); fp32 engines from the 24-bit stream.  Settable: an fp32 engine with this set draws the same
        # numbers as a bf16 one.
        self.inference_noise16 = dtype == BF16
        # Launches that sweep the batch last image to first (dvae_epilogue.reverse_walk: a scheduling hint, same results):
        # the first conv of each layer1 block starts where its producer has just finished writing, and the block's second
        # conv, sweeping forwards again, starts where the first one finished -- +1.2 % on the 2048-image purification step.
        # DVAE_REVERSE_WALK=<comma-separated conv names> overrides the set ("-" for none).
        rw = os.environ.get("DVAE_REVERSE_WALK")
        self._reverse_walk = frozenset(("layer1.0.conv1", "layer1.1.conv1") if rw is None else [n for n in rw.split(",") if n and n != "-"])
        self.probe = None   # (conv name, (start_event, stop_event)): bench.py times one conv launch with HIP events; cleared by that launch
        self._overlap = self.dev.type == "cuda"   # weight gradients on a side stream (see _wgrad)
        self._two_stream_decoders = os.environ.get("DVAE_TWO_STREAM_DECODERS", "1") != "0"

    # ------------------------------------------------------------------------------------------
    def _desc(self, sp: ConvSpec, b: int) -> ConvDesc:
        return ConvDesc(b, sp.hin, sp.hin, sp.cin, sp.cout, sp.k, sp.k, sp.stride, sp.pad, 1 if sp.transposed else 0)

    def desc(self, name, b):
        key = (name, b)
        d = self._descs.get(key)
        if d is None:
            d = self._descs[key] = self._desc(self.specs[name], b)
        return d

    def _alloc_workspaces(self):
        B = self.B
        wbytes = max(self.lib.dvae_conv_wgrad_workspace_bytes(C.byref(self._desc(sp, B))) for sp in self.specs.values())
        nb = 0
        for _, c in self.bns:
            rows = B * (self.S * self.S if c == 64 else self.h * self.h)
            nb = max(nb, self.lib.dvae_norm_workspace_bytes(1, rows, c))
        for c in (64, 32):
            nb = max(nb, self.lib.dvae_norm_workspace_bytes(B, self.S * self.S, c))   # monotone in the group count
        self.wgrad_ws = torch.empty(wbytes // 4 + 16, device=self.dev, dtype=torch.float32)
        # one workspace slice per conv as well: backward leaves every conv's weight-gradient slabs in its own slice and
        # reduces all of them in one launch (18 short, latency-bound reductions otherwise)
        self.wgrad_slice = {}
        off = 0
        for name, sp in self.specs.items():
            n = self.lib.dvae_conv_wgrad_workspace_bytes(C.byref(self._desc(sp, B))) // 4 + 16
            self.wgrad_slice[name] = (off, n)
            off += (n + 63) // 64 * 64
        self.wgrad_ws_all = torch.empty(off, device=self.dev, dtype=torch.float32)
        self._reduce_jobs = {}        # batch size -> (job table on the device, jobs, widest job)
        self._batch_reduce = True
        self.norm_ws = torch.empty(nb // 4 + 16, device=self.dev, dtype=torch.float32)

    def buf(self, name, shape, dtype=None):
        """Preallocated activation buffer (allocated on first use, then reused)."""
        t = self._bufs.get(name)
        if t is None:
            t = self._bufs[name] = torch.zeros(shape, device=self.dev, dtype=dtype or self.tdt)
        return t

    def act(self, name, hw, c, f32=False):
        cs = L.act_channels(c, F32 if f32 else self.dtype)
        return self.buf(name, (self.B, hw, hw, cs), torch.float32 if f32 else self.tdt)

    # ------------------------------------------------------------------------------------------
    def set_hyper(self, **kw):
        """Optimiser scalars live in device memory (graph replays read them there).  steps_done is
        advanced on the device by every adam_step; set it here only to (re)start a run."""
        idx = dict(lr=0, beta1=1, beta2=2, eps=3, max_norm=4, grad_scale=5)
        h = self._hyper_host
        for k, v in kw.items():
            if k == "steps_done":
                self.hyper[6:7].fill_(float(v))
            else:
                h[idx[k]] = float(v)
        self.hyper[:6].copy_(h[:6], non_blocking=True)

    def load_params(self, tensors: dict, strict=True):
        """Copy reference-named tensors (CPU or GPU) into the bucket / BatchNorm buffers."""
        seen = set()
        for n, s, o, k in self.layout:
            if n in tensors:
                self.param[n].copy_(tensors[n].reshape(s).to(self.dev, torch.float32))
                seen.add(n)
            elif strict:
                raise KeyError(n)
        for n in self.buffers:
            if n in tensors:
                self.buffers[n].copy_(tensors[n].to(self.dev, torch.float32))
                seen.add(n)
            elif strict:
                raise KeyError(n)
        self.invalidate()
        return seen

    def invalidate(self):
        """Parameters changed behind the engine's back: re-pack weights / re-fold BatchNorm lazily."""
        self.packed_valid = False
        self.fold_valid = False

    def _build_pack_jobs(self):
        nb = self.lib.dvae_conv_pack_job_bytes()
        host = torch.zeros(len(self.packed) * nb, dtype=torch.uint8)
        for i, ((name, op), t) in enumerate(self.packed.items()):
            check(self.lib.dvae_conv_pack_job(ptr(self.param[name + ".weight"]), ptr(t), C.byref(self.desc(name, 1)), op,
                                              self.dtype, host.data_ptr() + i * nb))
        self._pack_jobs = host.to(self.dev)

    def pack_weights(self):
        """Refresh every packed GEMM operand image (one launch) and the NHWC embedding table."""
        st = L.stream()
        if getattr(self, "_pack_jobs", None) is None:
            self._build_pack_jobs()
        check(self.lib.dvae_conv_pack_batch(ptr(self._pack_jobs), len(self.packed), self.dtype, st))
        check(self.lib.dvae_emb_pack(ptr(self.param["emb_p.weight"]), ptr(self.emb_table), self.cfg.num_classes, self.h * self.h,
                                     128, int(self.cfg.spatial_emb), self.dtype, st))
        self.packed_valid = True

    def fold_bn(self):
        st = L.stream()
        for n, c in self.bns:
            sc, sh = self.bn_fold[n]
            check(self.lib.dvae_bn_fold_eval(ptr(self.param[n + ".weight"]), ptr(self.param[n + ".bias"]),
                                             ptr(self.buffers[n + ".running_mean"]), ptr(self.buffers[n + ".running_var"]),
                                             ptr(sc), ptr(sh), c, BN_EPS, st))
        self.fold_valid = True

    # ------------------------------------------------------------------------------------------
    # primitive launches
    # ------------------------------------------------------------------------------------------
    def _conv(self, name, x, y, b, scale=None, shift=None, residual=None, relu=False, out_f32=False, in_ss=None, in_relu=False,
              out_stats=None, emb=None):
        """emb = (packed table, labels, second output): the conv also writes y + table[label] (decoder_p's input)."""
        et, el, eo = emb if emb is not None else (None, None, None)
        ep = Epilogue(ptr(scale), ptr(shift), ptr(residual), int(relu), int(out_f32), ptr(in_ss), int(in_relu), ptr(out_stats),
                      ptr(et), ptr(el), ptr(eo), int(emb is not None and not self.cfg.spatial_emb), int(name in getattr(self, "_reverse_walk", ())))
        pr = self.probe if (self.probe is not None and self.probe[0] == name) else None
        if pr:
            pr[1][0].record()
        check(self.lib.dvae_conv_fwd(ptr(x), ptr(self.packed[(name, L.OP_FWD)]), C.byref(ep), ptr(y),
                                     C.byref(self.desc(name, b)), self.dtype, L.stream()))
        if pr:
            pr[1][1].record()
            self.probe = None                              # one launch per request (an event pair costs its stream ~12 us)

    def _dgrad(self, name, dy, dx, b, accumulate=False):
        check(self.lib.dvae_conv_dgrad(ptr(dy), ptr(self.packed[(name, L.OP_DGRAD)]), ptr(dx), int(accumulate),
                                       C.byref(self.desc(name, b)), self.dtype, L.stream()))

    def _side_stream(self):
        st = getattr(self, "_side", None)
        if st is None:
            st = self._side = torch.cuda.Stream(device=self.dev)
        return st

    def _comm_stream(self):
        st = getattr(self, "_comm", None)
        if st is None:
            st = self._comm = torch.cuda.Stream(device=self.dev)
        return st

    def _wgrad(self, name, x, dy, b):
        """Weight (and bias) gradient of one conv, issued on the side stream: nothing later in backward reads it, so it
        runs next to the input-gradient chain (short, latency-bound launches) instead of in front of it.  All weight
        gradients share one workspace and one stream, so they stay ordered among themselves; backward() joins the
        side stream before it returns."""
        if not self._overlap:
            return self._wgrad_now(name, x, dy, b)
        side = self._side_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._wgrad_now(name, x, dy, b)

    def _early_names(self):
        """The convs whose weight gradients are complete before the stem's backward starts (the latent convs and both
        decoders).  Their parameters, with the class embedding, are the tail of the flat bucket from ``early_offset``."""
        return [n for n in self.specs if n in ("fc", "post_conv") or n.startswith("decoder")]

    @property
    def early_offset(self):
        return dict((n, o) for n, _, o, _ in self.layout)["fc_mu.weight"]

    def _reduce_table(self, b, part="all"):
        """Job records of the one-launch slab reduction for batch size b (built once per b).  ``part``: "all", or
        "early" / "late" = the two halves of a data-parallel step, whose first half is reduced (and its gradients sent on
        their way) while the stem's backward still runs."""
        ent = self._reduce_jobs.get((b, part))
        if ent is None:
            nb = self.lib.dvae_wgrad_reduce_job_bytes()
            early = self._early_names()
            names = [n for n in self.specs if part == "all" or (n in early) == (part == "early")]
            host = torch.zeros(len(names) * nb, dtype=torch.uint8)
            widest = 0
            for i, name in enumerate(names):
                off, n = self.wgrad_slice[name]
                db = self.grad[name + ".bias"] if self.specs[name].bias else None
                blocks = C.c_int(0)
                check(self.lib.dvae_wgrad_reduce_job(self.wgrad_ws_all.data_ptr() + 4 * off, ptr(self.grad[name + ".weight"]), ptr(db),
                                                     C.byref(self.desc(name, b)), self.dtype, host.data_ptr() + i * nb,
                                                     C.byref(blocks)))
                widest = max(widest, blocks.value)
            ent = self._reduce_jobs[(b, part)] = (host.to(self.dev), len(names), widest)
        return ent

    def _wgrad_reduce_all(self, b, part="all"):
        jobs, n, widest = self._reduce_table(b, part)
        check(self.lib.dvae_wgrad_reduce_batch(ptr(jobs), n, widest, L.stream()))

    def _wgrad_now(self, name, x, dy, b):
        sp = self.specs[name]
        if self._batch_reduce:
            off, n = self.wgrad_slice[name]
            check(self.lib.dvae_conv_wgrad_partial(ptr(x), ptr(dy), int(sp.bias), self.wgrad_ws_all.data_ptr() + 4 * off, n * 4,
                                                   C.byref(self.desc(name, b)), self.dtype, L.stream()))
            return
        db = self.grad[name + ".bias"] if sp.bias else None
        check(self.lib.dvae_conv_wgrad(ptr(x), ptr(dy), ptr(self.grad[name + ".weight"]), ptr(db), ptr(self.wgrad_ws),
                                       self.wgrad_ws.numel() * 4, C.byref(self.desc(name, b)), self.dtype, L.stream()))

    def _bn_train(self, name, x, y, rows, c, residual=None, relu=True):
        mean, invstd = self.bn_save[name]
        check(self.lib.dvae_norm_fwd_train(ptr(x), ptr(residual), ptr(y), ptr(self.param[name + ".weight"]),
                                           ptr(self.param[name + ".bias"]), ptr(self.buffers[name + ".running_mean"]),
                                           ptr(self.buffers[name + ".running_var"]), ptr(mean), ptr(invstd),
                                           ptr(self.norm_ws), self.norm_ws.numel() * 4, 1, rows, c, int(relu),
                                           BN_MOMENTUM, BN_EPS, self.dtype, L.stream()))

    def _bn_bwd(self, name, dy, y, x, dx, dres, rows, c, relu=True):
        mean, invstd = self.bn_save[name]
        check(self.lib.dvae_norm_bwd(ptr(dy), ptr(y), ptr(x), ptr(dx), ptr(dres), ptr(self.param[name + ".weight"]),
                                     ptr(mean), ptr(invstd), ptr(self.grad[name + ".weight"]), ptr(self.grad[name + ".bias"]),
                                     ptr(self.norm_ws), self.norm_ws.numel() * 4, 1, rows, c, int(relu), self.dtype,
                                     L.stream()))

    def _in_fwd(self, tag, x, y, b, c):
        mean = self.buf(tag + ".mean", (self.B, c), torch.float32)
        invstd = self.buf(tag + ".invstd", (self.B, c), torch.float32)
        check(self.lib.dvae_norm_fwd_train(ptr(x), None, ptr(y), None, None, None, None, ptr(mean), ptr(invstd),
                                           ptr(self.norm_ws), self.norm_ws.numel() * 4, b, self.S * self.S, c, 1,
                                           0.0, IN_EPS, self.dtype, L.stream()))

    def _in_bwd(self, tag, dy, y, x, dx, b, c):
        mean = self.buf(tag + ".mean", (self.B, c), torch.float32)
        invstd = self.buf(tag + ".invstd", (self.B, c), torch.float32)
        check(self.lib.dvae_norm_bwd(ptr(dy), ptr(y), ptr(x), ptr(dx), None, None, ptr(mean), ptr(invstd), None, None,
                                     ptr(self.norm_ws), self.norm_ws.numel() * 4, b, self.S * self.S, c, 1, self.dtype,
                                     L.stream()))

    # ------------------------------------------------------------------------------------------
    # input / output conversion
    # ------------------------------------------------------------------------------------------
    def input_from_nchw(self, x_nchw, name="x0"):
        b = x_nchw.shape[0]
        x0 = self.act(name, self.S, 3)
        check(self.lib.dvae_nchw_to_act(ptr(x_nchw.contiguous()), ptr(x0), b, 3, self.S, self.S, self.dtype, L.stream()))
        return x0

    def input_from_u8(self, data_u8, first, b, index=None, name="x0"):
        x0 = self.act(name, self.S, 3)
        check(self.lib.dvae_u8_to_act(ptr(data_u8), ptr(index), int(first), ptr(x0), b, self.S * self.S, self.dtype, L.stream()))
        return x0

    def eps_from_nchw(self, eps_nchw):
        b = eps_nchw.shape[0]
        e = self.buf("eps", (self.B, self.h, self.h, self.Ld), torch.float32)
        check(self.lib.dvae_nchw_to_act(ptr(eps_nchw.contiguous()), ptr(e), b, self.Ld, self.h, self.h, F32, L.stream()))
        return e

    def eps_philox(self, b, first=0, index=None, stream_id=0):
        """Noise specification for forward(): drawn in-kernel (no eps tensor round trip through HBM)."""
        return PhiloxEps(int(first), index, int(stream_id))

    def eps_philox_tensor(self, b, first=0, index=None, stream_id=0):
        """The same noise materialised as an fp32 act tensor (tests, debugging)."""
        e = self.buf("eps", (self.B, self.h, self.h, self.Ld), torch.float32)
        check(self.lib.dvae_philox_normal(ptr(e), ptr(index), int(first), b, self.h * self.h * self.Ld, self.seed,
                                          int(stream_id), L.stream()))
        return e

    def to_nchw(self, act_t, b, c, hw, coff=0, out=None):
        if out is None:
            out = torch.empty(b, c, hw, hw, device=self.dev, dtype=torch.float32)
        check(self.lib.dvae_act_to_nchw(ptr(act_t), ptr(out), b, c, hw, hw, act_t.shape[-1], coff,
                                        int(act_t.dtype == torch.float32), self.dtype, L.stream()))
        return out

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward(self, x0, y, eps, b, train: bool, need_rec: bool = True, need_mulv: bool | None = None):
        """x0: input act [B,S,S,4]; y: int64 [b]; eps: fp32 act [B,h,h,L].  Leaves rec_pre / p (and
        every intermediate needed by backward) in the engine's buffers and returns them.
        ``need_rec=False`` (inference only) leaves the reconstruction branch out and returns (None, p): for callers
        that read nothing but the perturbation of this pass (main_vae_disentangle.py:61-62).
        ``need_mulv`` (default: training only): mu | log_var are written to the "mulv" buffer (fp32; "mulv.act" in
        the act dtype for bf16 inference with in-kernel noise).  Without it the inference head may keep them in
        registers, and those buffers then hold whatever an earlier call left."""
        stem_out = self.stem(x0, b, train)
        _, _, z = self.latent(stem_out, eps, b, keep_eps=train, need_mulv=train if need_mulv is None else need_mulv, y=y)
        if isinstance(eps, PhiloxEps):
            eps = self.buf("eps", (self.B, self.h, self.h, self.Ld), torch.float32) if train else None
        if (not train) and need_rec and self._two_stream_decoders and self._fuse_in and self._fuse_stats and self.dev.type == "cuda":
            # The two decoders of an inference forward are independent chains of six launches each (on the fused route
            # they share no workspace): decoder_p runs on the side stream, so that a launch of one chain fills the CUs
            # the other chain's launch has already left and the gaps between its dependent launches.
            side, cur = self._side_stream(), torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                p = self.perturb(stem_out, z, y, b, train)
            rec_pre = self._decoder("decoder", z, b, train)
            cur.wait_stream(side)
        else:
            rec_pre = self._decoder("decoder", z, b, train) if (train or need_rec) else None
            p = self.perturb(stem_out, z, y, b, train)
        self._last = dict(b=b, x0=x0, y=y, eps=eps, stem_out=stem_out, train=train)
        return rec_pre, p

    def stem(self, x0, b, train: bool):
        """resnet.py:2139-2150.  Train mode: conv -> batch-stat BatchNorm(+residual)+ReLU kernels.  Eval
        mode: BatchNorm folded into the conv epilogue (scale/shift + residual + ReLU)."""
        if not self.packed_valid:
            self.pack_weights()
        if not train and not self.fold_valid:
            self.fold_bn()
        S = self.S
        rows_S = b * S * S
        a = self.act("bn1.out", S, 64)
        if train:
            c0 = self.act("conv1.out", S, 64)
            self._conv("conv1", x0, c0, b)
            self._bn_train("bn1", c0, a, rows_S, 64)
        else:
            sc, sh = self.bn_fold["bn1"]
            self._conv("conv1", x0, a, b, scale=sc, shift=sh, relu=True)
        for blk in self.blocks:
            sp1 = self.specs[blk + ".conv1"]
            ho, co = sp1.hout, sp1.cout
            rows = b * ho * ho
            has_sc = (blk + ".shortcut.0") in self.specs
            t = self.act(blk + ".bn1.out", ho, co)
            out = self.act(blk + ".out", ho, co)
            if train:
                c1 = self.act(blk + ".conv1.out", ho, co)
                self._conv(blk + ".conv1", a, c1, b)
                self._bn_train(blk + ".bn1", c1, t, rows, co)
                c2 = self.act(blk + ".conv2.out", ho, co)
                self._conv(blk + ".conv2", t, c2, b)
                if has_sc:
                    cs = self.act(blk + ".shortcut.0.out", ho, co)
                    self._conv(blk + ".shortcut.0", a, cs, b)
                    s = self.act(blk + ".shortcut.1.out", ho, co)
                    self._bn_train(blk + ".shortcut.1", cs, s, rows, co, relu=False)
                else:
                    s = a
                self._bn_train(blk + ".bn2", c2, out, rows, co, residual=s)
            else:
                sc, sh = self.bn_fold[blk + ".bn1"]
                self._conv(blk + ".conv1", a, t, b, scale=sc, shift=sh, relu=True)
                if has_sc:
                    s = self.act(blk + ".shortcut.1.out", ho, co)
                    sc, sh = self.bn_fold[blk + ".shortcut.1"]
                    self._conv(blk + ".shortcut.0", a, s, b, scale=sc, shift=sh)
                else:
                    s = a
                sc, sh = self.bn_fold[blk + ".bn2"]
                self._conv(blk + ".conv2", t, out, b, scale=sc, shift=sh, residual=s, relu=True)
            a = out
        return a

    def latent(self, stem_out, eps, b, keep_eps=True, need_mulv=True, y=None):
        """encode (fc_mu|fc_var as one GEMM) -> reparameterise (+ KL partial sums) -> post_conv.
        resnet.py:2197-2199, :2245-2246, :2257-2259.  With labels y (and where the conv supports it) post_conv also
        writes decoder_p's input z + emb_p(y) (resnet.py:2212, :2219); perturb() then skips its own pass."""
        emb = None
        self._emb_done = False
        if y is not None and self._fuse_emb:
            emb = (self.emb_table, y, self.act("dec_p.in", self.h, 128))
            self._emb_done = True
        if not self.packed_valid:
            self.pack_weights()
        h = self.h
        # Inference with in-kernel noise takes no loss from mu / log_var: they stay in the act dtype there
        # (halves the largest tensor of the forward pass on the bf16 path).  Training keeps them fp32.
        lean = isinstance(eps, PhiloxEps) and not keep_eps and self.dtype == BF16
        z1 = self.act("z1", h, self.Ld)
        if lean and self._fuse_head_post and not need_mulv:
            # one launch for fc, reparameterisation and post_conv: mu, log_var and z1 never exist in memory (z1's buffer
            # keeps whatever an earlier call left)
            z = self.act("z", h, 128)
            et, el, eo = emb if emb is not None else (None, None, None)
            check(self.lib.dvae_fc_reparam_post_fwd_philox(
                ptr(stem_out), ptr(self.packed[("fc", L.OP_FWD)]), ptr(self.param["fc.bias"]),
                ptr(self.packed[("post_conv", L.OP_FWD)]), ptr(self.param["post_conv.bias"]), ptr(z), ptr(et), ptr(el), ptr(eo),
                int(emb is not None and not self.cfg.spatial_emb), C.byref(self.desc("fc", b)), C.byref(self.desc("post_conv", b)),
                self.Ld, ptr(eps.index), eps.first, self.seed, eps.stream_id, self.dtype, L.stream()))
            return None, z1, z
        if lean and self._fuse_head and not need_mulv:
            # one launch: mu / log_var never leave the registers of the GEMM that makes them
            check(self.lib.dvae_fc_reparam_fwd_philox(ptr(stem_out), ptr(self.packed[("fc", L.OP_FWD)]), ptr(self.param["fc.bias"]),
                                                      ptr(z1), C.byref(self.desc("fc", b)), self.Ld, ptr(eps.index), eps.first,
                                                      self.seed, eps.stream_id, self.dtype, L.stream()))
            z = self.act("z", h, 128)
            self._conv("post_conv", z1, z, b, shift=self.param["post_conv.bias"], emb=emb)
            return None, z1, z
        mulv = self.act("mulv.act", h, 2 * self.Ld) if lean else self.act("mulv", h, 2 * self.Ld, f32=True)
        self._conv("fc", stem_out, mulv, b, shift=self.param["fc.bias"], out_f32=not lean)
        if isinstance(eps, PhiloxEps):
            eps_out = self.buf("eps", (self.B, h, h, self.Ld), torch.float32) if keep_eps else None
            check(self.lib.dvae_reparam_kl_fwd_philox(ptr(mulv), int(lean), ptr(eps_out), ptr(z1), ptr(self.kl_partial), b, h * h,
                                                      self.Ld, ptr(eps.index), eps.first, self.seed, eps.stream_id, self.dtype,
                                                      L.stream()))
        else:
            check(self.lib.dvae_reparam_kl_fwd(ptr(mulv), ptr(eps), ptr(z1), ptr(self.kl_partial), b * h * h, self.Ld,
                                               self.dtype, L.stream()))
        z = self.act("z", h, 128)
        self._conv("post_conv", z1, z, b, shift=self.param["post_conv.bias"], emb=emb)
        return mulv, z1, z

    def perturb(self, stem_out, z, y, b, train=True):
        """decoder_p(Embedding + stem_out | Embedding + z | Embedding), resnet.py:2211-2221."""
        cfg, h = self.cfg, self.h
        ep = self.act("dec_p.in", h, 128)
        if getattr(self, "_emb_done", False):            # post_conv has written it already
            self._emb_done = False
            return self._decoder("decoder_p", ep, b, train)
        add = None
        if cfg.sample_wise:
            add = stem_out if cfg.use_y else z
        check(self.lib.dvae_emb_add_fwd(ptr(self.emb_table), ptr(y), ptr(add), ptr(ep), b, h * h, 128,
                                        int(cfg.spatial_emb), self.dtype, L.stream()))
        return self._decoder("decoder_p", ep, b, train)

    def _decoder(self, d, zin, b, train=True):
        """ConvT(k4,s2) -> ConvT(k3) -> IN+ReLU -> ConvT(k3) -> IN+ReLU -> ConvT(k3)  (resnet.py:2011-2050).
        Training keeps every intermediate for backward.  Inference (where the consuming conv supports it)
        computes only the InstanceNorm statistics and lets the next conv normalise its input while staging
        it: the normalised maps d3 / d5 are never written."""
        S = self.S
        d1 = self.act(d + ".d1", S, 64)
        self._conv(d + "1.0", zin, d1, b, shift=self.param[d + "1.0.bias"])
        d2 = self.act(d + ".d2", S, 64)
        d4 = self.act(d + ".d4", S, 32)
        d6 = self.act(d + ".d6", S, 3, f32=True)       # rec_pre / perturbation stay fp32 on every path
        fuse = (not train) and self._fuse_in
        if fuse and self._fuse_stats:
            # statistics ride on the producing conv's store, normalisation on the consuming conv's load
            ch1 = self.buf(d + ".in1.chunks", (self.B, 8, 2, d2.shape[-1]), torch.float32)
            ch2 = self.buf(d + ".in2.chunks", (self.B, 8, 2, d4.shape[-1]), torch.float32)
            self._conv(d + "1.1", d1, d2, b, shift=self.param[d + "1.1.bias"], out_stats=ch1)
            ss1 = self._in_stats_chunks(d + ".in1", ch1, b, 64)
            self._conv(d + "2.0", d2, d4, b, shift=self.param[d + "2.0.bias"], in_ss=ss1, in_relu=True, out_stats=ch2)
            ss2 = self._in_stats_chunks(d + ".in2", ch2, b, 32)
            self._conv(d + "3.0", d4, d6, b, shift=self.param[d + "3.0.bias"], out_f32=True, in_ss=ss2, in_relu=True)
            return d6
        self._conv(d + "1.1", d1, d2, b, shift=self.param[d + "1.1.bias"])
        if fuse:
            ss1 = self._in_stats(d + ".in1", d2, b, 64)
            self._conv(d + "2.0", d2, d4, b, shift=self.param[d + "2.0.bias"], in_ss=ss1, in_relu=True)
            ss2 = self._in_stats(d + ".in2", d4, b, 32)
            self._conv(d + "3.0", d4, d6, b, shift=self.param[d + "3.0.bias"], out_f32=True, in_ss=ss2, in_relu=True)
            return d6
        d3 = self.act(d + ".d3", S, 64)
        self._in_fwd(d + ".in1", d2, d3, b, 64)
        self._conv(d + "2.0", d3, d4, b, shift=self.param[d + "2.0.bias"])
        d5 = self.act(d + ".d5", S, 32)
        self._in_fwd(d + ".in2", d4, d5, b, 32)
        self._conv(d + "3.0", d5, d6, b, shift=self.param[d + "3.0.bias"], out_f32=True)
        return d6

    def _in_stats(self, tag, x, b, c):
        mean = self.buf(tag + ".mean", (self.B, c), torch.float32)
        invstd = self.buf(tag + ".invstd", (self.B, c), torch.float32)
        ss = self.buf(tag + ".ss", (self.B, 2, c), torch.float32)
        check(self.lib.dvae_norm_fwd_stats(ptr(x), None, None, ptr(mean), ptr(invstd), ptr(ss), ptr(self.norm_ws),
                                           self.norm_ws.numel() * 4, b, self.S * self.S, c, IN_EPS, self.dtype, L.stream()))
        return ss

    def _in_stats_chunks(self, tag, chunks, b, c):
        """InstanceNorm statistics from the (sum, sum of squares) chunks the producing conv left behind."""
        mean = self.buf(tag + ".mean", (self.B, c), torch.float32)
        invstd = self.buf(tag + ".invstd", (self.B, c), torch.float32)
        ss = self.buf(tag + ".ss", (self.B, 2, c), torch.float32)
        check(self.lib.dvae_norm_stats_from_chunks(ptr(chunks), self.S * self.S // 8, None, None, ptr(mean), ptr(invstd), ptr(ss), b,
                                                   self.S * self.S, c, IN_EPS, self.dtype, L.stream()))
        return ss

    # ------------------------------------------------------------------------------------------
    # losses
    # ------------------------------------------------------------------------------------------
    def loss_sums(self):
        """rec = sigmoid(rec_pre) and the four loss sums of this batch into self.sums (device)."""
        b = self._last["b"]
        S = self.S
        rec = self.buf("rec", (self.B, S, S, 4), torch.float32)
        check(self.lib.dvae_recon_loss_fwd(ptr(self.act("decoder.d6", S, 3, f32=True)), ptr(self.act("decoder_p.d6", S, 3, f32=True)),
                                           ptr(self._last["x0"]), ptr(rec), ptr(self.recon_partial), b * S * S,
                                           self.dtype, L.stream()))
        check(self.lib.dvae_loss_reduce(ptr(self.recon_partial), ptr(self.kl_partial), ptr(self.sums), L.stream()))
        return rec

    def loss_gate(self, epoch: int, kld_target: float, kld_weight: float, world: int = 1):
        """sums (already summed over ranks when world > 1) -> scalars and gradient coefficients."""
        b = self._last["b"]
        n_loc = b * 3 * self.S * self.S
        r_loc = b * self.h * self.h
        check(self.lib.dvae_loss_gate(ptr(self.sums), ptr(self.scal), float(n_loc * world), float(r_loc * world),
                                      float(n_loc), float(r_loc), kld_target, kld_weight, int(epoch > 10), L.stream()))

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def backward(self, d_rec_pre=None, d_p=None, d_mulv_extra=None, allreduce=None):
        """Gradients of every parameter on the path into self.G.  With no arguments the loss
        gradients come from the fused loss kernels (scal must hold the gate coefficients).
        ``allreduce(tensor)`` (data-parallel steps): the gradient bucket is summed over ranks in two pieces.  The tail of
        the bucket -- latent convs, decoders, class embedding: two thirds of it -- is complete once the weight gradient
        of fc_mu|fc_var has run; it is reduced and sent from the side stream while the main stream still walks the
        stem's backward.  The head follows when backward has joined the side stream."""
        L_ = self._last
        b, S, h = L_["b"], self.S, self.h
        rows_h = b * h * h
        st = L.stream
        if d_rec_pre is None:
            d_rec_pre = self.act("g.decoder.d6", S, 3)
            d_p = self.act("g.decoder_p.d6", S, 3)
            check(self.lib.dvae_recon_loss_bwd(ptr(self.buf("rec", (self.B, S, S, 4), torch.float32)),
                                               ptr(self.act("decoder_p.d6", S, 3, f32=True)), ptr(L_["x0"]), ptr(self.scal),
                                               ptr(d_rec_pre), ptr(d_p), b * S * S, self.dtype, st()))
        cfg = self.cfg
        z = self.act("z", h, 128)
        # decoder_p first: its input gradient feeds the embedding and (sample-wise) z or stem_out
        d_ep = self.act("g.dec_p.in", h, 128)
        self._decoder_bwd("decoder_p", self.act("dec_p.in", h, 128), d_p, d_ep, b, accumulate=False)
        check(self.lib.dvae_emb_bwd(ptr(d_ep), ptr(L_["y"]), ptr(self.grad["emb_p.weight"]), b, h * h, 128,
                                    cfg.num_classes, int(cfg.spatial_emb), self.dtype, st()))
        if cfg.sample_wise and not cfg.use_y:
            d_z = d_ep
            self._decoder_bwd("decoder", z, d_rec_pre, d_z, b, accumulate=True)
        else:
            d_z = self.act("g.z", h, 128)
            self._decoder_bwd("decoder", z, d_rec_pre, d_z, b, accumulate=False)
        # post_conv
        z1 = self.act("z1", h, self.Ld)
        self._wgrad("post_conv", z1, d_z, b)
        d_z1 = self.act("g.z1", h, self.Ld)
        self._dgrad("post_conv", d_z, d_z1, b)
        # reparameterise + KL
        d_mulv = self.act("g.mulv", h, 2 * self.Ld)
        check(self.lib.dvae_reparam_kl_bwd(ptr(d_z1), ptr(self.act("mulv", h, 2 * self.Ld, f32=True)), ptr(L_["eps"]),
                                           ptr(self.scal), ptr(d_mulv), rows_h, self.Ld, self.dtype, st()))
        if d_mulv_extra is not None:
            d_mulv.add_(d_mulv_extra)
        stem_out = L_["stem_out"]
        self._wgrad("fc", stem_out, d_mulv, b)
        # The slabs of the latent convs and the decoders are reduced as soon as the last of them is there, on the side
        # stream under the stem's backward: only the stem's own slabs are left for the reduction at the end.
        split = self._batch_reduce and self._overlap and getattr(self, "_split_reduce", True)
        if split:
            # everything behind early_offset is final once the side stream gets here (emb_bwd ran on the main stream
            # before the fc weight gradient was issued, and _wgrad made the side stream wait for that point)
            side = self._side_stream()
            with torch.cuda.stream(side):
                self._wgrad_reduce_all(b, "early")
            if allreduce is not None:
                comm = self._comm_stream()
                comm.wait_stream(side)                       # a stream of its own: the stem's weight gradients, which
                with torch.cuda.stream(comm):                # follow on the side stream, do not queue behind the collective
                    allreduce(self.G[self.early_offset:])
        if cfg.sample_wise and cfg.use_y:
            d_stem = d_ep
            self._dgrad("fc", d_mulv, d_stem, b, accumulate=True)
        else:
            d_stem = self.act("g.stem_out", h, 128)
            self._dgrad("fc", d_mulv, d_stem, b)
        # stem, in reverse
        d_out = d_stem
        for i in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[i]
            a_in = self.act(self.blocks[i - 1] + ".out", *self._block_out_shape(self.blocks[i - 1])) if i > 0 else self.act("bn1.out", S, 64)
            d_out = self._block_bwd(blk, a_in, d_out, b)
        d_c0 = self.act("g.conv1.out", S, 64)
        self._bn_bwd("bn1", d_out, self.act("bn1.out", S, 64), self.act("conv1.out", S, 64), d_c0, None, b * S * S, 64)
        self._wgrad("conv1", L_["x0"], d_c0, b)
        if self._batch_reduce:                 # every conv of the path has left its slabs: one reduction for all
            if self._overlap:
                side = self._side_stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._wgrad_reduce_all(b, "late" if split else "all")
            else:
                self._wgrad_reduce_all(b)
        if self._overlap:
            torch.cuda.current_stream().wait_stream(self._side_stream())
        if allreduce is not None:
            allreduce(self.G[:self.early_offset] if split else self.G)
            if split:
                torch.cuda.current_stream().wait_stream(self._comm_stream())

    def _block_out_shape(self, blk):
        sp = self.specs[blk + ".conv1"]
        return sp.hout, sp.cout

    def _block_bwd(self, blk, a_in, d_out, b):
        sp1 = self.specs[blk + ".conv1"]
        ho, co, hi, ci = sp1.hout, sp1.cout, sp1.hin, sp1.cin
        rows = b * ho * ho
        has_sc = (blk + ".shortcut.0") in self.specs
        out = self.act(blk + ".out", ho, co)
        t = self.act(blk + ".bn1.out", ho, co)
        c1 = self.act(blk + ".conv1.out", ho, co)
        c2 = self.act(blk + ".conv2.out", ho, co)
        d_in = self.act("g." + blk + ".in", hi, ci)
        d_c2 = self.act("g." + blk + ".conv2.out", ho, co)
        if has_sc:
            gm = self.act("g." + blk + ".res", ho, co)
            self._bn_bwd(blk + ".bn2", d_out, out, c2, d_c2, gm, rows, co)
        else:
            self._bn_bwd(blk + ".bn2", d_out, out, c2, d_c2, d_in, rows, co)   # d_in <- masked grad (identity shortcut)
        self._wgrad(blk + ".conv2", t, d_c2, b)
        d_t = self.act("g." + blk + ".bn1.out", ho, co)
        self._dgrad(blk + ".conv2", d_c2, d_t, b)
        d_c1 = self.act("g." + blk + ".conv1.out", ho, co)
        self._bn_bwd(blk + ".bn1", d_t, t, c1, d_c1, None, rows, co)
        self._wgrad(blk + ".conv1", a_in, d_c1, b)
        self._dgrad(blk + ".conv1", d_c1, d_in, b, accumulate=not has_sc)
        if has_sc:
            cs = self.act(blk + ".shortcut.0.out", ho, co)
            d_cs = self.act("g." + blk + ".shortcut.0.out", ho, co)
            self._bn_bwd(blk + ".shortcut.1", gm, None, cs, d_cs, None, rows, co, relu=False)
            self._wgrad(blk + ".shortcut.0", a_in, d_cs, b)
            self._dgrad(blk + ".shortcut.0", d_cs, d_in, b, accumulate=True)
        return d_in

    def _decoder_bwd(self, d, zin, d_d6, d_zin, b, accumulate):
        S = self.S
        d1, d2, d3 = self.act(d + ".d1", S, 64), self.act(d + ".d2", S, 64), self.act(d + ".d3", S, 64)
        d4, d5 = self.act(d + ".d4", S, 32), self.act(d + ".d5", S, 32)
        self._wgrad(d + "3.0", d5, d_d6, b)
        g5 = self.act("g." + d + ".d5", S, 32)
        self._dgrad(d + "3.0", d_d6, g5, b)
        g4 = self.act("g." + d + ".d4", S, 32)
        self._in_bwd(d + ".in2", g5, d5, d4, g4, b, 32)
        self._wgrad(d + "2.0", d3, g4, b)
        g3 = self.act("g." + d + ".d3", S, 64)
        self._dgrad(d + "2.0", g4, g3, b)
        g2 = self.act("g." + d + ".d2", S, 64)
        self._in_bwd(d + ".in1", g3, d3, d2, g2, b, 64)
        self._wgrad(d + "1.1", d1, g2, b)
        g1 = self.act("g." + d + ".d1", S, 64)
        self._dgrad(d + "1.1", g2, g1, b)
        self._wgrad(d + "1.0", zin, g1, b)
        self._dgrad(d + "1.0", g1, d_zin, b, accumulate=accumulate)

    # ------------------------------------------------------------------------------------------
    # optimiser
    # ------------------------------------------------------------------------------------------
    def adam_step(self):
        """clip_grad_norm_(max_norm) + Adam over the flat bucket, then refresh the packed weights."""
        st = L.stream()
        check(self.lib.dvae_sqnorm_partial(ptr(self.G), self.nparam, ptr(self.norm_partial), st))
        check(self.lib.dvae_clip_adam(ptr(self.P), ptr(self.G), ptr(self.M), ptr(self.V), self.nparam, ptr(self.hyper),
                                      ptr(self.norm_partial), ptr(self.scal), st))
        self.pack_weights()
        self.fold_valid = False

    # ------------------------------------------------------------------------------------------
    # whole steps
    # ------------------------------------------------------------------------------------------
    def train_step(self, x0, y, eps, b, epoch, kld_target, kld_weight, allreduce=None, world=1):
        """forward + losses + backward + clip + Adam for one batch already resident as act tensors.
        ``allreduce(tensor)`` sums a tensor over ranks in place (None on one GPU)."""
        self.forward(x0, y, eps, b, True)
        self.loss_sums()
        if allreduce is not None:
            allreduce(self.sums)
        self.loss_gate(epoch, kld_target, kld_weight, world)
        self.backward(allreduce=allreduce)
        self.adam_step()

    def train_step_graphed(self, x0, y, eps, b, epoch, kld_target, kld_weight, allreduce=None, world=1):
        """train_step() replayed from captured HIP graphs: a step is ~200 short launches, and at the reference's batch
        size the host cannot issue them as fast as the GPU retires them.  Everything a step reads lives in fixed
        device buffers (activations, hyper-parameters, the Adam step count), so a capture stays valid from step to
        step; the per-step values -- labels and the noise counters -- are copied into fixed buffers before the replay.
        Between ranks the collectives stay outside the graphs (three segments: forward+loss sums | gate+backward |
        clip+Adam).  The first call with a given (batch, gate, constants) runs eagerly, the second captures."""
        if not isinstance(eps, PhiloxEps):
            raise ValueError("train_step_graphed draws its noise in-kernel: pass eps_philox(...)")
        if x0.data_ptr() != self.act("x0", self.S, 3).data_ptr():
            raise ValueError("train_step_graphed reads the engine's own input buffer: build x0 with input_from_*()")
        gy = self.buf("graph.y", (self.B,), torch.int64)
        gi = self.buf("graph.index", (self.B,), torch.int32)
        gy[:b].copy_(y)
        if eps.index is not None:
            gi[:b].copy_(eps.index[:b])
        else:
            torch.arange(eps.first, eps.first + b, device=self.dev, dtype=torch.int32, out=gi[:b])
        geps = PhiloxEps(0, gi, eps.stream_id)
        key = (b, epoch > 10, float(kld_target), float(kld_weight), world, eps.stream_id, allreduce is not None)
        graphs = getattr(self, "_graphs", None)
        if graphs is None:
            graphs = self._graphs = {}
        if not self.packed_valid:
            self.pack_weights()
        entry = graphs.get(key)
        if entry is None:
            graphs[key] = "warm"
            self.train_step(x0, gy, geps, b, epoch, kld_target, kld_weight, allreduce, world)
            return
        if entry == "warm":
            segs = [lambda: (self.forward(x0, gy, geps, b, True), self.loss_sums()),
                    lambda: (self.loss_gate(epoch, kld_target, kld_weight, world), self.backward()),
                    self.adam_step]
            if allreduce is None:
                segs = [lambda s=segs: [f() for f in s]]
            entry = []
            for f in segs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    f()
                entry.append(g)
            graphs[key] = entry
        if allreduce is None:
            entry[0].replay()
        else:
            entry[0].replay()
            allreduce(self.sums)
            entry[1].replay()
            allreduce(self.G)
            entry[2].replay()
        self.packed_valid, self.fold_valid = True, False

    def purify_batch(self, data_u8, out_u8, noise_out, labels, first, b, stage, eps_nchw=None, global_first=None,
                     skip_unread=False):
        """Purify images [first, first+b) of the resident uint8 set (main_vae_disentangle.py:59-69).
        Writes uint8 HWC into out_u8[first:first+b] and the (last) perturbation, NCHW fp32, into
        noise_out[first:first+b].  ``eps_nchw``: optional (pass0, pass1) injected noise; otherwise the
        noise is Philox keyed by the global image index ``global_first + i`` (shard-invariant).
        ``skip_unread``: the first pass computes only what :61-62 reads of it (the perturbation); the reference also
        runs that pass's reconstruction decoder and drops the result.  Outputs are bit-identical either way."""
        S = self.S
        gf = first if global_first is None else global_first
        y = labels[first:first + b]
        x0 = self.input_from_u8(data_u8, first, b)
        n16 = L.PHILOX_STREAM_16 if self.inference_noise16 else 0
        eps = self.eps_from_nchw(eps_nchw[0]) if eps_nchw is not None else self.eps_philox(b, gf, stream_id=1 | n16)
        _, p = self.forward(x0, y, eps, b, False, need_rec=not skip_unread)
        if stage == 0:
            check(self.lib.dvae_purify_quant(ptr(x0), ptr(p), 0, None, ptr(out_u8), int(first), b, S * S, self.dtype, L.stream()))
        else:
            x1 = self.act("x1", S, 3)
            check(self.lib.dvae_purify_quant(ptr(x0), ptr(p), 0, ptr(x1), None, 0, b, S * S, self.dtype, L.stream()))
            eps = self.eps_from_nchw(eps_nchw[1]) if eps_nchw is not None else self.eps_philox(b, gf, stream_id=2 | n16)
            rec_pre, p = self.forward(x1, y, eps, b, False)
            # rec = sigmoid(rec_pre), clamp and quantise in one launch
            check(self.lib.dvae_purify_quant(ptr(rec_pre), None, 2, None, ptr(out_u8), int(first), b, S * S, self.dtype, L.stream()))
        if noise_out is not None:
            self.to_nchw(p, b, 3, S, out=noise_out[first:first + b])
