"""The CIFAR ResNet-18 trunk (conv1 .. layer4) of the classifier stage on libdvae_hip.

``main_vae_disentangle.py:97-125`` trains ``ResNet18`` (/root/reference/configs/models/resnet.py:62-122) on the purified
set for 60 epochs x 390 steps: in a two-stage run that is the longest phase.  ``ResNetTrunkEngine`` runs its convolutional
part -- every conv, BatchNorm, residual add and ReLU, forward and backward -- on the library's kernels (the same
``dvae_conv_*`` / ``dvae_norm_*`` entry points as the D-VAE's stem, plus the shapes only the classifier has: 3x3
128->256 s2, 256->256 at 8x8, 256->512 s2, 512->512 at 4x4 and their 1x1 shortcuts).  Pooling, the linear layer and the
cross-entropy stay stock torch ops on a [B,512,4,4] tensor.

The engine borrows the launch helpers and the block schedule of ``DVAEEngine`` (``stem`` / ``_block_bwd`` walk whatever
``blocks`` / ``specs`` name); what differs is where parameters live: the classifier's ``nn.Parameter`` tensors are used in
place (any torch optimiser steps them), gradients are views of one flat fp32 bucket.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from ._lib import F32, check, ptr
from .engine import ConvSpec, DVAEEngine

WIDTHS = ((64, 1), (128, 2), (256, 2), (512, 2))


def trunk_specs(num_blocks, S):
    """(block names, conv specs keyed by the reference's module names) of ResNet(BasicBlock, num_blocks) at S x S."""
    specs = {"conv1": ConvSpec("conv1", 3, 64, 3, 1, 1, False, S, False)}
    blocks = []
    cin, h = 64, S
    for i, (c, s) in enumerate(WIDTHS):
        for j in range(num_blocks[i]):
            blk = f"layer{i + 1}.{j}"
            stride = s if j == 0 else 1
            blocks.append(blk)
            specs[blk + ".conv1"] = ConvSpec(blk + ".conv1", cin, c, 3, stride, 1, False, h, False)
            specs[blk + ".conv2"] = ConvSpec(blk + ".conv2", c, c, 3, 1, 1, False, h // stride, False)
            if stride != 1 or cin != c:
                specs[blk + ".shortcut.0"] = ConvSpec(blk + ".shortcut.0", cin, c, 1, stride, 0, False, h, False)
            cin, h = c, h // stride
    return blocks, specs


class ResNetTrunkEngine(DVAEEngine):
    """Static schedule of the trunk for batches of up to ``max_batch`` images.  ``params`` / ``buffers``: the module's
    tensors by reference name (fp32, on the device); they are read in place at every call."""

    def __init__(self, num_blocks, params: dict, buffers: dict, max_batch: int, device, dtype: int = F32, image_size: int = 32):
        self.B, self.dtype, self.dev = int(max_batch), dtype, torch.device(device)
        self.lib = L.lib()
        self.tdt = torch.float32 if dtype == F32 else torch.bfloat16
        self.S = image_size
        self.blocks, self.specs = trunk_specs(num_blocks, image_size)
        self.out_hw = self.specs[self.blocks[-1] + ".conv2"].hout
        self.out_c = self.specs[self.blocks[-1] + ".conv2"].cout
        self.bns = [("bn1", 64)]
        for blk in self.blocks:
            c = self.specs[blk + ".conv1"].cout
            self.bns += [(blk + ".bn1", c), (blk + ".bn2", c)]
            if (blk + ".shortcut.0") in self.specs:
                self.bns.append((blk + ".shortcut.1", c))
        self.param, self.buffers = params, buffers
        f32 = dict(device=self.dev, dtype=torch.float32)
        # ---- gradient bucket: one view per trainable tensor of the trunk, 16-byte-aligned slots ----
        names = []
        for n in self.specs:
            names.append(n + ".weight")
        for n, _ in self.bns:
            names += [n + ".weight", n + ".bias"]
        self.names = names
        off, slots = 0, []
        for n in names:
            k = params[n].numel()
            slots.append((n, off, k))
            off += (k + 3) // 4 * 4
        self.G = torch.zeros(off, **f32)
        self.grad = {n: self.G[o:o + k].view(params[n].shape) for n, o, k in slots}
        # ---- BatchNorm scratch ----
        tot = sum(c for _, c in self.bns)
        self.FOLD = torch.zeros(2 * tot, **f32)
        self.bn_fold, o = {}, 0
        for n, c in self.bns:
            self.bn_fold[n] = (self.FOLD[o:o + c], self.FOLD[tot + o:tot + o + c])
            o += c
        self.bn_save = {n: (torch.zeros(c, **f32), torch.zeros(c, **f32)) for n, c in self.bns}
        # ---- packed weights (forward image; input-gradient image for every conv but the first) ----
        self.packed = {}
        for name, sp in self.specs.items():
            d = self._desc(sp, 1)
            for op in (L.OP_FWD, L.OP_DGRAD):
                if name == "conv1" and op == L.OP_DGRAD:
                    continue
                self.packed[(name, op)] = torch.zeros(self.lib.dvae_conv_packed_elems(C.byref(d), op), device=self.dev, dtype=self.tdt)
        self._pack_jobs = None
        self._pack_ptrs = None
        # ---- workspaces ----
        self._bufs, self._descs = {}, {}
        wbytes = max(self.lib.dvae_conv_wgrad_workspace_bytes(C.byref(self._desc(sp, self.B))) for sp in self.specs.values())
        self.wgrad_ws = torch.empty(wbytes // 4 + 16, device=self.dev, dtype=torch.float32)
        nb = 0
        for n, c in self.bns:
            sp = self.specs.get(n.replace(".bn1", ".conv1").replace(".bn2", ".conv2").replace(".shortcut.1", ".shortcut.0"),
                                self.specs["conv1"])
            nb = max(nb, self.lib.dvae_norm_workspace_bytes(1, self.B * sp.hout * sp.hout, c))
        self.norm_ws = torch.empty(nb // 4 + 16, device=self.dev, dtype=torch.float32)
        self._batch_reduce = False                       # every weight gradient reduces its own slabs (dvae_conv_wgrad)
        self._overlap = self.dev.type == "cuda"          # ... on the side stream, beside the input-gradient chain
        self.packed_valid = self.fold_valid = False
        self.probe = None
        self._last = None

    # ------------------------------------------------------------------------------------------
    def pack_weights(self):
        """Every conv's packed GEMM images re-made from the fp32 parameters in one launch.  The job table holds the
        parameters' addresses: it is rebuilt if a parameter's storage moved."""
        ptrs = tuple(self.param[n + ".weight"].data_ptr() for n in self.specs)
        if self._pack_jobs is None or ptrs != self._pack_ptrs:
            self._build_pack_jobs()
            self._pack_ptrs = ptrs
        check(self.lib.dvae_conv_pack_batch(ptr(self._pack_jobs), len(self.packed), self.dtype, L.stream()))
        self.packed_valid = True

    def forward_trunk(self, x_nchw, train: bool):
        """[b,3,S,S] fp32 in [0,1] -> layer4's output as fp32 NCHW [b,C,h,h]; keeps what backward needs when ``train``."""
        b = x_nchw.shape[0]
        if b > self.B:
            raise ValueError(f"batch {b} exceeds the engine's {self.B}")
        x0 = self.input_from_nchw(x_nchw.float())
        out = self.stem(x0, b, train)
        self._last = dict(b=b, x0=x0, train=train)
        return self.to_nchw(out, b, self.out_c, self.out_hw)

    def backward_trunk(self, d_out_nchw):
        """Gradient of layer4's output (fp32 NCHW) -> every trunk parameter's gradient in the bucket."""
        L_ = self._last
        if L_ is None or not L_["train"]:
            raise RuntimeError("backward_trunk needs a preceding training-mode forward_trunk")
        b, S = L_["b"], self.S
        last = self.blocks[-1]
        d_out = self.act("g." + last + ".out", self.out_hw, self.out_c)
        check(self.lib.dvae_nchw_to_act(ptr(d_out_nchw.contiguous().float()), ptr(d_out), b, self.out_c, self.out_hw, self.out_hw,
                                        self.dtype, L.stream()))
        for i in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[i]
            a_in = self.act(self.blocks[i - 1] + ".out", *self._block_out_shape(self.blocks[i - 1])) if i > 0 else self.act("bn1.out", S, 64)
            d_out = self._block_bwd(blk, a_in, d_out, b)
        d_c0 = self.act("g.conv1.out", S, 64)
        self._bn_bwd("bn1", d_out, self.act("bn1.out", S, 64), self.act("conv1.out", S, 64), d_c0, None, b * S * S, 64)
        self._wgrad("conv1", L_["x0"], d_c0, b)
        if self._overlap:
            torch.cuda.current_stream().wait_stream(self._side_stream())
