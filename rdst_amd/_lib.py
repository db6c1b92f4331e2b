"""ctypes binding of librdst_hip.so (the C ABI of include/rdst_hip.h).  Fails loudly: there is no
fallback when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDST_HIP_LIB: load another build of the same C ABI (tools/ use it for the -DRDST_DEBUG library); the default
# is the in-tree release library, and a missing file raises either way
LIB_PATH = os.environ.get("RDST_HIP_LIB") or os.path.join(_HERE, "librdst_hip.so")

F32, BF16, F32X3 = 0, 1, 2
EINVAL, ENOTSUP = -10001, -10002
ACT_NONE, ACT_GELU, ACT_LEAKY02, ACT_LEAKY001 = 0, 1, 2, 3

_p, _i, _l, _f, _z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/rdst_hip.h declares
SIGNATURES = {
    "rdst_abi_version": (_i, []),
    "rdst_last_error": (C.c_char_p, []),
    "rdst_wattn_fwd": (_i, [_p, _l, _p, _p, _i, _p, _l, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "rdst_wattn_bwd_workspace": (_z, [_i, _i, _i, _i, _i, _i]),
    "rdst_wattn_bwd": (_i, [_p, _l, _p, _p, _i, _p, _l, _p, _l, _p, _p, _z, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "rdst_ln_linear_fwd_workspace": (_z, [_i, _i]),
    "rdst_ln_linear_fwd_workspace2": (_z, [_i, _i, _i]),
    "rdst_ln_linear_fwd": (_i, [_p, _l, _p, _p, _i, _p, _p, _p, _l, _p, _l, _p, _p, _z, _l, _i, _i, _f, _i, _p]),
    "rdst_ln_linear_bwd_workspace": (_z, [_l, _i, _i]),
    "rdst_ln_linear_bwd": (_i, [_p, _l, _p, _p, _p, _i, _p, _p, _l, _p, _l, _p, _l, _p, _p, _p, _p, _p, _z,
                                _l, _i, _i, _f, _i, _p]),
    "rdst_ln_linear_bwd2": (_i, [_p, _l, _p, _p, _p, _i, _p, _p, _l, _p, _l, _p, _l, _p, _p, _p, _p, _p, _z,
                                 _l, _i, _i, _f, _i, _p, _p, _l]),
    "rdst_mlp_fused_supported": (_i, [_i, _i, _i]),
    "rdst_mlp_fwd_workspace": (_z, [_i, _i]),
    "rdst_mlp_fwd_packable": (_i, [_i, _i, _i]),
    "rdst_mlp_fwd": (_i, [_p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _z, _l, _i, _i, _i, _p]),
    "rdst_mlp_bwd_workspace": (_z, [_l, _i, _i]),
    "rdst_mlp_bwd": (_i, [_p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _l, _p, _p, _p, _p, _p, _p, _p, _z, _l, _i, _i, _i,
                          _p]),
    "rdst_wattn_fwd_lse": (_i, [_p, _l, _p, _p, _l, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "rdst_wattn_bwd_lse": (_i, [_p, _l, _p, _p, _l, _p, _l, _p, _p, _l, _p, _p, _z, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "rdst_swin_attn_fwd_supported": (_i, [_i, _i, _i, _i]),
    "rdst_swin_attn_fwd_workspace": (_z, [_i]),
    "rdst_swin_attn_fwd": (_i, [_p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _l, _p, _l, _p, _p, _z, _i, _i, _i, _i, _i, _i, _i,
                                _f, _i, _p]),
    "rdst_conv_fwd_workspace": (_z, [_i, _i, _i]),
    "rdst_conv_fwd_workspace2": (_z, [_i, _i, _i, _i]),
    "rdst_conv_fwd": (_i, [_p, _l, _i, _p, _p, _p, _l, _p, _l, _p, _z, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "rdst_conv_bwd_workspace": (_z, [_i, _i, _i, _i, _i, _i]),
    "rdst_conv_bwd": (_i, [_p, _l, _i, _p, _p, _l, _p, _l, _p, _l, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i,
                           _f, _i, _i, _p]),
    "rdst_pack_batch": (_i, [_p, _i, _p]),
    "rdst_ln_linear_fwd_packable": (_i, [_i, _i, _i, _i, _i, _i]),
    "rdst_conv_fwd_packable": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "rdst_reduce_batch_begin": (_i, []),
    "rdst_reduce_batch_end": (_i, [_p]),
    "rdst_reduce_batch_abort": (_i, []),
    "rdst_wattn_fwd_drop": (_i, [_p, _l, _p, _p, _i, _p, _l, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _p, _p]),
    "rdst_wattn_bwd_drop": (_i, [_p, _l, _p, _p, _i, _p, _l, _p, _l, _p, _p, _z, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _p, _p]),
    "rdst_wattn_drop_mask": (_i, [_p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "rdst_nchw_to_rows": (_i, [_p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "rdst_rows_to_nchw": (_i, [_p, _l, _p, _i, _i, _i, _i, _i, _p]),
    "rdst_upsample2_fwd": (_i, [_p, _l, _p, _l, _i, _i, _i, _i, _i, _p]),
    "rdst_upsample2_bwd": (_i, [_p, _l, _p, _l, _i, _i, _i, _i, _i, _p]),
    "rdst_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _l, _p]),
    # the seg-UNet of the perceptual loss (ABI v5)
    "rdst_u_scratch_bytes": (_z, []),
    "rdst_u_conv": (_i, [_p, _l, _i, _i, _p, _l, _i, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p,
                         C.POINTER(C.c_int)]),
    "rdst_u_bn_stats_from": (_i, [_p, _i, _l, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p]),
    "rdst_u_stem_fwd": (_i, [_p, _p, _p, _l, _i, _i, _i, _i, _i, _p]),
    "rdst_u_stem_dgrad": (_i, [_p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "rdst_u_bn_stats": (_i, [_p, _l, _l, _i, _p, _p, _f, _f, _p, _p, _p, _p, _i, _p]),
    "rdst_u_bn_apply": (_i, [_p, _l, _p, _p, _l, _p, _p, _l, _i, _p, _l, _l, _i, _i, _p]),
    "rdst_u_bn_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _p, _l, _p, _l, _p, _l, _l, _i, _p, _i, _p]),
    "rdst_u_maxpool_fwd": (_i, [_p, _l, _p, _l, _p, _i, _i, _i, _i, _i, _p]),
    "rdst_u_maxpool_bwd": (_i, [_p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p]),
    "rdst_u_sumpool2": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p]),
    "rdst_u_pair_loss_fwd": (_i, [_p, _l, _p, _l, _l, _i, _i, _f, _i, _p, _p, _i, _p]),
    "rdst_u_pair_loss_bwd": (_i, [_p, _l, _p, _l, _l, _i, _i, _f, _p, _p, _l, _p, _l, _i, _p]),
    "rdst_u_dice_fwd": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _f, _f, _i, _p, _p, _p, _i, _p]),
    "rdst_u_dice_bwd": (_i, [_p, _l, _p, _l, _p, _l, _i, _p, _p, _p, _l, _i, _i, _p]),
}

ABI_VERSION = 11             # must equal rdst_abi_version() of the loaded library (argument lists change between versions)
PREPACKED = (1 << 64) - 1   # RDST_PREPACKED ((size_t)-1)


class PackJob(C.Structure):   # rdst_pack_job
    _fields_ = [("kind", _i), ("W", _p), ("gamma", _p), ("beta", _p), ("bias", _p), ("out", _p), ("N", _i), ("K", _i), ("s", _f)]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library or raise: the product path never runs without it."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"rdst_amd: {LIB_PATH} is missing. Build it with `python -m rdst_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:  # a declared symbol the .so does not export: calling it raises
            setattr(lib, name, _missing(name))
            continue
        fn.restype = res
        fn.argtypes = args
    got = lib.rdst_abi_version()
    if got != ABI_VERSION:   # a stale library would be called with shifted arguments (sizes read as pointers)
        raise RuntimeError(f"rdst_amd: {LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: "
                           "rebuild it with `python -m rdst_amd.build --force` (and `--debug` for the _dbg library)")
    _lib = lib
    return lib


def loaded() -> bool:
    return _lib is not None


def _missing(name):
    def raiser(*_a, **_k):
        raise RuntimeError(f"rdst_amd: {LIB_PATH} does not export {name}; rebuild with `python -m rdst_amd.build --force`")
    return raiser


class HipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().rdst_last_error().decode(errors="replace")
        raise HipError(f"{what} failed (rc={rc}): {msg}")
