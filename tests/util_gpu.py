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


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
