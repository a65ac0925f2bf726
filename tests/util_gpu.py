"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

import dvae_amd
from dvae_amd import _lib as L


def to_act(t_nchw, dtype=L.F32):
    """CPU NCHW fp32 -> device NHWC act tensor with padded channels."""
    b, c, h, w = t_nchw.shape
    cs = L.act_channels(c, dtype)
    out = torch.zeros(b, h, w, cs, dtype=torch.float32)
    out[..., :c] = t_nchw.permute(0, 2, 3, 1)
    return out.to("cuda", torch.float32 if dtype == L.F32 else torch.bfloat16).contiguous()


def from_act(a, c):
    """device NHWC act -> CPU NCHW fp32 (first c channels)."""
    return a[..., :c].float().permute(0, 3, 1, 2).contiguous().cpu()


def _np(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().double().numpy()
    return np.asarray(a, dtype=np.float64)


def rel_err(a, b):
    a, b = _np(a), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def l2_err(a, b):
    """||a - b||_2 / ||b||_2"""
    a, b = _np(a), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))
