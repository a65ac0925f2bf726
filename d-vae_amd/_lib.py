"""ctypes binding of libdvae_hip.so (C ABI declared in include/dvae_hip.h).

The library is built in-tree by ``make`` / ``__graft_entry__.build()``.  There is no fallback: if the
shared object is missing, every product entry point raises.
"""
import ctypes as C
import os

import torch

F32, BF16 = 0, 1
OP_FWD, OP_DGRAD = 0, 1
NPARTIAL = 1024
PHILOX_STREAM_16 = 0x80000000     # stream ids with this bit draw eight normals per Philox call from 16-bit uniforms
SCAL_COUNT = 16
SCAL = dict(loss=0, loss_rec=1, loss_rec_p=2, loss_p=3, kld=4, gate_kl=5, gate_p=6, grad_norm=7,
            c_rec=8, c_p=9, c_kl=10)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdvae_hip.so")


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "H", "W", "Cin", "Cout", "kh", "kw", "stride", "pad", "transposed")]


class Epilogue(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p),
                ("relu", C.c_int32), ("out_f32", C.c_int32), ("in_scale_shift", C.c_void_p), ("in_relu", C.c_int32),
                ("out_stats", C.c_void_p), ("emb_table", C.c_void_p), ("emb_labels", C.c_void_p), ("emb_out", C.c_void_p),
                ("emb_plain", C.c_int32), ("reverse_walk", C.c_int32)]


_P, _I, _L, _F, _D, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
_SIGS = {
    "dvae_act_channels": (C.c_int, [_I]),
    "dvae_set_option": (C.c_int, [C.c_char_p, _I]),
    "dvae_u8_to_act": (C.c_int, [_P, _P, _L, _P, _I, _I, _I, _P]),
    "dvae_nchw_to_act": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dvae_act_to_nchw": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dvae_conv_packed_elems": (_Z, [C.POINTER(ConvDesc), _I]),
    "dvae_conv_pack": (C.c_int, [_P, _P, C.POINTER(ConvDesc), _I, _I, _P]),
    "dvae_conv_pack_job_bytes": (_Z, []),
    "dvae_conv_pack_job": (C.c_int, [_P, _P, C.POINTER(ConvDesc), _I, _I, _P]),
    "dvae_conv_pack_batch": (C.c_int, [_P, _I, _I, _P]),
    "dvae_conv_fwd": (C.c_int, [_P, _P, C.POINTER(Epilogue), _P, C.POINTER(ConvDesc), _I, _P]),
    "dvae_conv_supports_input_norm": (C.c_int, [C.POINTER(ConvDesc), _I]),
    "dvae_conv_supports_emb_out": (C.c_int, [C.POINTER(ConvDesc), _I]),
    "dvae_norm_fwd_stats": (C.c_int, [_P] * 7 + [_Z, _I, _I, _I, _F, _I, _P]),
    "dvae_conv_dgrad": (C.c_int, [_P, _P, _P, _I, C.POINTER(ConvDesc), _I, _P]),
    "dvae_conv_wgrad_workspace_bytes": (_Z, [C.POINTER(ConvDesc)]),
    "dvae_conv_wgrad": (C.c_int, [_P, _P, _P, _P, _P, _Z, C.POINTER(ConvDesc), _I, _P]),
    "dvae_conv_wgrad_uses_big_tiles": (C.c_int, [C.POINTER(ConvDesc), _I]),
    "dvae_conv_wgrad_partial": (C.c_int, [_P, _P, _I, _P, _Z, C.POINTER(ConvDesc), _I, _P]),
    "dvae_wgrad_reduce_job_bytes": (_Z, []),
    "dvae_wgrad_reduce_job": (C.c_int, [_P, _P, _P, C.POINTER(ConvDesc), _I, _P, C.POINTER(C.c_int)]),
    "dvae_wgrad_reduce_batch": (C.c_int, [_P, _I, _I, _P]),
    "dvae_norm_workspace_bytes": (_Z, [_I, _I, _I]),
    "dvae_norm_fwd_train": (C.c_int, [_P] * 9 + [_P, _Z, _I, _I, _I, _I, _F, _F, _I, _P]),
    "dvae_norm_fwd_train_static": (C.c_int, [_P] * 9 + [_I, _I, _I, _I, _F, _F, _I, _P]),
    "dvae_bn_fwd_eval": (C.c_int, [_P] * 8 + [_I, _I, _I, _F, _I, _P]),
    "dvae_bn_fold_eval": (C.c_int, [_P] * 6 + [_I, _F, _P]),
    "dvae_norm_bwd": (C.c_int, [_P] * 10 + [_P, _Z, _I, _I, _I, _I, _I, _P]),
    "dvae_emb_pack": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dvae_emb_add_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dvae_emb_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dvae_philox_normal": (C.c_int, [_P, _P, _L, _I, _I, C.c_uint64, C.c_uint32, _P]),
    "dvae_reparam_kl_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dvae_conv_out_stats_floats": (_Z, [C.POINTER(ConvDesc), _I]),
    "dvae_norm_stats_from_chunks": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "dvae_fc_reparam_supported": (C.c_int, [C.POINTER(ConvDesc), _I, _I]),
    "dvae_fc_reparam_fwd_philox": (C.c_int, [_P, _P, _P, _P, C.POINTER(ConvDesc), _I, _P, _L, C.c_uint64, C.c_uint32, _I, _P]),
    "dvae_fc_reparam_post_supported": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _I, _I]),
    "dvae_fc_reparam_post_fwd_philox": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, C.POINTER(ConvDesc), C.POINTER(ConvDesc),
                                                   _I, _P, _L, C.c_uint64, C.c_uint32, _I, _P]),
    "dvae_reparam_kl_fwd_philox": (C.c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _P, _L, C.c_uint64, C.c_uint32, _I, _P]),
    "dvae_reparam_kl_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dvae_recon_loss_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "dvae_loss_reduce": (C.c_int, [_P, _P, _P, _P]),
    "dvae_loss_gate": (C.c_int, [_P, _P, _D, _D, _D, _D, _F, _F, _I, _P]),
    "dvae_recon_loss_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dvae_purify_quant": (C.c_int, [_P, _P, _I, _P, _P, _L, _I, _I, _I, _P]),
    "dvae_poison_u8": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "dvae_sqnorm_partial": (C.c_int, [_P, _L, _P, _P]),
    "dvae_clip_adam": (C.c_int, [_P, _P, _P, _P, _L, _P, _P, _P, _P]),
}
EXPORTS = ["dvae_version", "dvae_last_error"] + list(_SIGS)

_lib = None


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()) first; "
                               "there is no fallback path")
        l = C.CDLL(LIB_PATH)
        l.dvae_version.restype = C.c_char_p
        l.dvae_last_error.restype = C.c_char_p
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libdvae_hip: error {rc}: {lib().dvae_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of the calling thread's current stream on the current device.  Through torch's C helpers where they
    exist: `torch.cuda.current_stream().cuda_stream` builds a Python Stream object on every call, ~3 us, and every launch
    asks for it -- a third of the host time of a 125-launch train step, which is bound by its launch rate."""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _RAW_STREAM(_CUR_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def act_channels(c, dtype=F32):
    p = 4
    while p < c:
        p <<= 1
    return max(p, 8) if dtype == BF16 else p
