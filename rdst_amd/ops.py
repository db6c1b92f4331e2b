"""Autograd bindings of the HIP kernels (the host side above the C ABI of include/rdst_hip.h).

Every function here takes CUDA(=HIP) tensors, enqueues hand-written gfx950 kernels from
librdst_hip.so on torch's current stream, and raises if the library is missing or a tensor lives
on the CPU.  torch is used for device memory, streams and autograd bookkeeping only.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_LEAKY02, ACT_NONE, BF16, F32  # noqa: F401


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"rdst_amd: unsupported activation dtype {t.dtype} (float32 or bfloat16)")


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "rdst_amd: the HIP path needs GPU tensors; there is no CPU fallback "
                "(the CPU oracle lives in oracle/ and is test-only)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _rows(t: torch.Tensor) -> tuple[torch.Tensor, int]:
    """View `t` (..., C) as rows with one leading dimension `ld` (elements); copy only if the
    layout cannot be expressed that way."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    if t.dim() == 1:
        return t, t.shape[0]
    ld = t.stride(-2)
    ok = ld >= t.shape[-1]
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            ok = False
            break
        exp *= t.shape[d]
    if not ok:
        t = t.contiguous()
        ld = t.shape[-1]
    return t, ld


def _f32p(t):
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous()
    return t.data_ptr()


# ------------------------------------------------------------------------------------------------
# K1 / K2: fused window attention
# ------------------------------------------------------------------------------------------------
class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, table, H, W, heads, ws, shift, scale):
        _need_gpu(qkv, table)
        lib = _lib.load()
        B = qkv.numel() // (H * W * qkv.shape[-1])
        C = qkv.shape[-1] // 3
        qkv_r, ld = _rows(qkv)
        tab = table.detach().float().contiguous()
        out = torch.empty(qkv.shape[:-1] + (C,), dtype=qkv.dtype, device=qkv.device)
        _lib.check(lib.rdst_wattn_fwd(qkv_r.data_ptr(), ld, tab.data_ptr(), out.data_ptr(), C, B, H, W, C, heads,
                                      ws, shift, float(scale), _dtype_code(qkv), _stream()), "rdst_wattn_fwd")
        ctx.save_for_backward(qkv_r, tab)
        ctx.geom = (B, H, W, C, heads, ws, shift, float(scale), ld)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, tab = ctx.saved_tensors
        B, H, W, C, heads, ws, shift, scale, ld = ctx.geom
        lib = _lib.load()
        dout_r, ldd = _rows(dout)
        dqkv = torch.empty(qkv.shape[:-1] + (3 * C,), dtype=qkv.dtype, device=qkv.device)
        dtable = torch.empty_like(tab)
        nbytes = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
        wsp = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=qkv.device)
        _lib.check(lib.rdst_wattn_bwd(qkv.data_ptr(), ld, tab.data_ptr(), dout_r.data_ptr(), ldd, dqkv.data_ptr(),
                                      3 * C, dtable.data_ptr(), wsp.data_ptr(), nbytes, B, H, W, C, heads, ws, shift,
                                      scale, _dtype_code(qkv), _stream()), "rdst_wattn_bwd")
        return dqkv, dtable, None, None, None, None, None, None


def window_attention(qkv: torch.Tensor, table: torch.Tensor, H: int, W: int, heads: int, ws: int, shift: int,
                     scale: float) -> torch.Tensor:
    """Fused roll + window partition + (q*scale)k^T + relative-position bias + shift mask + softmax
    + @v + window reverse + un-roll on token-major qkv (..., 3C) -> (..., C).
    Replaces networks/swin_transformer_sr.py:244-267 with :117-138 inside (minus the Linears)."""
    return _WindowAttention.apply(qkv, table, H, W, heads, ws, shift, scale)
