"""Autograd bindings of the HIP kernels (the host side above the C ABI of include/rdst_hip.h).

Every function here takes CUDA(=HIP) tensors, enqueues hand-written gfx950 kernels from
librdst_hip.so on torch's current stream, and raises if the library is missing or a tensor lives
on the CPU.  torch is used for device memory, streams and autograd bookkeeping only.

Activations are "rows": (..., C) tensors whose last dim is contiguous and whose leading dims collapse
to one row stride (so a slice ``buf[..., :C]`` of a wider buffer is passed without a copy).
Parameters are always fp32; activations are fp32 (parity mode) or bf16 (throughput mode).
"""
from __future__ import annotations

import os
import threading
import weakref

from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_LEAKY001, ACT_LEAKY02, ACT_NONE, BF16, F32, F32X3  # noqa: F401


# The "fp32x3" compute mode (set_compute_dtype("fp32x3") on a network): fp32 tensors everywhere, but the GEMM-shaped kernels
# take their operands as two bf16 terms (16 mantissa bits) and run on the bf16 matrix cores (RDST_F32X3, csrc/mfma.h:
# Mma<float, true>) — the parity mode that holds the 4-decimal PSNR bar at a fraction of the exact-fp32 cost.
# The mode belongs to the MODULE: a network keeps ``compute_code`` (F32 / F32X3 / BF16) next to ``compute_dtype`` and its
# forward runs inside ``compute_scope(code)``; every autograd Function reads the code ONCE, in its forward, and keeps it in
# its ctx, so a backward never looks at the scope or at the default below (two networks in different modes can interleave
# their forwards and backwards freely).  F32_SPLIT is only the DEFAULT for fp32 tensors outside any scope (op-level calls).
F32_SPLIT = False
_scope_tls = threading.local()


def set_f32_split(on: bool) -> None:
    """Default arithmetic of fp32 ops called OUTSIDE a network forward (op-level tests, tools); networks carry their own."""
    global F32_SPLIT
    F32_SPLIT = bool(on)


class compute_scope:
    """``with ops.compute_scope(code):`` — the fp32 arithmetic (F32 exact / F32X3 split) of every op called inside, on this
    thread; ``None`` leaves the enclosing scope (or the default) in force.  Scopes nest."""

    def __init__(self, code: Optional[int]):
        if code not in (None, F32, _lib.F32X3, BF16):
            raise ValueError(f"rdst_amd: bad compute code {code!r}")
        self.code = code

    def __enter__(self):
        self.prev = getattr(_scope_tls, "code", None)
        if self.code is not None:
            _scope_tls.code = self.code
        return self

    def __exit__(self, *exc):
        _scope_tls.code = self.prev
        return False


def resolve_compute_dtype(dtype):
    """set_compute_dtype's argument -> (activation dtype, compute code): torch.float32 / 'fp32' = exact parity mode,
    'fp32x3' = fp32 tensors with split-bf16 GEMMs, torch.bfloat16 / 'bf16' = throughput mode."""
    if dtype == "fp32x3":
        return torch.float32, _lib.F32X3
    if dtype in ("fp32", torch.float32):
        return torch.float32, F32
    if dtype in ("bf16", torch.bfloat16):
        return torch.bfloat16, BF16
    raise ValueError("compute dtype must be torch.float32 ('fp32'), 'fp32x3' or torch.bfloat16 ('bf16')")


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        sc = getattr(_scope_tls, "code", None)
        if sc is None or sc == BF16:   # (a bf16 network's fp32 side tensors: the image boundary ops, exact)
            return _lib.F32X3 if (sc is None and F32_SPLIT) else F32
        return sc
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"rdst_amd: unsupported activation dtype {t.dtype} (float32 or bfloat16)")


def _elt_code(t: torch.Tensor) -> int:
    """Element type alone (layout-only ops: no arithmetic mode to choose)."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"rdst_amd: unsupported activation dtype {t.dtype} (float32 or bfloat16)")


def _need_gpu(*ts: Optional[torch.Tensor]) -> None:
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
    if t.dim() == 1:
        t = t.unsqueeze(0)
    C = t.shape[-1]
    if C != 1 and t.stride(-1) != 1:
        t = t.contiguous()
    ld = None
    exp = 0
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] == 1:
            continue
        if ld is None:
            ld = t.stride(d)
            exp = ld * t.shape[d]
        elif t.stride(d) != exp:
            return t.contiguous(), C
        else:
            exp *= t.shape[d]
    if ld is None:
        ld = C
    if ld < C:
        return t.contiguous(), C
    return t, ld


def _param(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _grad_like(param: torch.Tensor) -> torch.Tensor:
    """Destination of a parameter gradient: the flat-bucket view when rdst_amd.dp offers one (the kernel then
    writes straight into the all-reduce / Adam buffer), else a fresh tensor."""
    from . import dp
    v = dp.take_grad_view(param)
    if v is not None and v.shape == param.shape and v.dtype == param.dtype and v.device == param.device:
        return v
    return _fresh_grad(torch.empty_like(param))


def _fresh_grad(t: torch.Tensor) -> torch.Tensor:
    """A parameter-gradient destination that is NOT a bucket view: autograd may read it (``p.grad += t``) as soon as the
    node that returns it has returned, so its deferred reduction must have run by then (_ReduceBatch.settle)."""
    _ReduceBatch.fresh += 1
    return t


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# Prepacked weights: one rdst_pack_batch per network forward instead of a pack kernel in front of every op
# ------------------------------------------------------------------------------------------------
class PackPlan:
    """The bf16 fragment images of all the weights a network's forward reads (rdst_pack_batch, include/rdst_hip.h).
    Discovered on one forward (every op reports the pack it just did for itself), then refreshed as a whole at the start
    of each later forward; an op that finds its weights in the plan passes the image with workspace_bytes = PREPACKED.
    A plan is only used while every parameter of the owning module still has the dtype and lives at the address it had
    when the plan was recorded (``valid(owner)``): re-pointing ``p.data`` (FlatAdam's flat buffer), ``.to()`` or ``.half()``
    drops the plan and the next forward rediscovers it.  Plans live in a module-level WeakKeyDictionary keyed by the
    owner, never in the module's ``__dict__``: ``copy.deepcopy(net)`` / ``torch.save(net)`` neither see nor share them."""

    def __init__(self):
        self.keys, self.specs, self.tensors = {}, [], []
        self.arena = None
        self.jobs = None
        self.misses = 0
        self.owner_sig = None

    @staticmethod
    def _key(kind, w, lw, lb, b, N, K, s):
        return (kind, w.data_ptr(), _ptr(lw) or 0, _ptr(lb) or 0, _ptr(b) or 0, N, K, float(s))

    def record(self, kind, w, lw, lb, b, N, K, s, nbytes, key=None):
        k = key if key is not None else self._key(kind, w, lw, lb, b, N, K, s)
        if k not in self.keys:
            self.keys[k] = len(self.specs)
            self.specs.append((kind, N, K, float(s), int(nbytes)))
            self.tensors.append((w, lw, lb, b))

    def record_mlp(self, w1, lw, lb, b1, w2, b2, C, hid, nb1, nb2):
        """Two consecutive Linear images (fc1 with the LayerNorm folded in, fc2) = the workspace layout of rdst_mlp_fwd."""
        k = ("mlp", w1.data_ptr(), w2.data_ptr(), _ptr(lw) or 0, _ptr(lb) or 0, _ptr(b1) or 0, _ptr(b2) or 0)
        if k not in self.keys:
            self.record(PACK_LINEAR, w1, lw, lb, b1, hid, C, 1.0, nb1, key=k)
            self.record(PACK_LINEAR, w2, None, None, b2, C, hid, 1.0, (nb2 + 255) // 256 * 256, key=k + ("fc2",))

    def lookup_mlp(self, w1, lw, lb, b1, w2, b2):
        k = ("mlp", w1.data_ptr(), w2.data_ptr(), _ptr(lw) or 0, _ptr(lb) or 0, _ptr(b1) or 0, _ptr(b2) or 0)
        i = self.keys.get(k)
        if i is None:
            self.misses += 1
            return None
        return self.arena.data_ptr() + self.offsets[i]

    def record_swinattn(self, wq, lw, lb, bq, wp, bp, C, nbq, nbp):
        """The workspace layout of rdst_swin_attn_fwd: the sectioned qkv image (norm1 folded in), then the proj image."""
        k = ("swinattn", wq.data_ptr(), wp.data_ptr(), _ptr(lw) or 0, _ptr(lb) or 0, _ptr(bq) or 0, _ptr(bp) or 0)
        if k not in self.keys:
            self.record(PACK_LINEAR_SEC3, wq, lw, lb, bq, 3 * C, C, 1.0, nbq, key=k)
            self.record(PACK_LINEAR, wp, None, None, bp, C, C, 1.0, (nbp + 255) // 256 * 256, key=k + ("proj",))

    def lookup_swinattn(self, wq, lw, lb, bq, wp, bp):
        k = ("swinattn", wq.data_ptr(), wp.data_ptr(), _ptr(lw) or 0, _ptr(lb) or 0, _ptr(bq) or 0, _ptr(bp) or 0)
        i = self.keys.get(k)
        if i is None:
            self.misses += 1
            return None
        return self.arena.data_ptr() + self.offsets[i]

    @staticmethod
    def signature(owner):
        return (getattr(owner, "compute_dtype", None), getattr(owner, "compute_code", None)) + tuple(
            (p.data_ptr(), p.dtype) for p in owner.parameters())

    def finalize(self, device, owner=None):
        if not self.specs:
            return False
        self.owner_sig = self.signature(owner) if owner is not None else None
        total = sum(sp[4] for sp in self.specs)
        self.arena = torch.empty(total, dtype=torch.uint8, device=device)
        self.jobs = (_lib.PackJob * len(self.specs))()
        self.offsets, off = [], 0
        for i, ((kind, N, K, s, nb), (w, lw, lb, b)) in enumerate(zip(self.specs, self.tensors)):
            j = self.jobs[i]
            j.kind, j.W, j.gamma, j.beta, j.bias = kind, w.data_ptr(), _ptr(lw), _ptr(lb), _ptr(b)
            j.out, j.N, j.K, j.s = self.arena.data_ptr() + off, N, K, s
            self.offsets.append(off)
            off += nb
        self.ptrs = [tuple(_ptr(t) or 0 for t in ts) for ts in self.tensors]
        return True

    def valid(self, owner=None):
        """The LIVE parameters of the owner are where (and what) they were when the plan was recorded."""
        if owner is not None and self.owner_sig is not None and self.signature(owner) != self.owner_sig:
            return False
        return all(tuple(_ptr(t) or 0 for t in ts) == p for ts, p in zip(self.tensors, self.ptrs))

    def run(self):
        import ctypes
        _lib.check(_lib.load().rdst_pack_batch(ctypes.cast(self.jobs, ctypes.c_void_p), len(self.specs), _stream()),
                   "rdst_pack_batch")

    def lookup(self, kind, w, lw, lb, b, N, K, s):
        i = self.keys.get(self._key(kind, w, lw, lb, b, N, K, s))
        if i is None:
            self.misses += 1
            return None
        return self.arena.data_ptr() + self.offsets[i]


_plan_active: Optional[PackPlan] = None
_plan_recording: Optional[PackPlan] = None
_PLANS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()   # owner module -> PackPlan


def pack_plan_of(owner) -> Optional[PackPlan]:
    return _PLANS.get(owner)
PACK_LINEAR, PACK_CONV3_FWD, PACK_LINEAR_SEC3, PACK_LINEAR_X3, PACK_CONV3_FWD_X3 = 0, 1, 2, 3, 4


def _linear_pack_kind(code: int) -> int:
    """The packed image a Linear forward of this compute code reads: bf16 fragments, or hi / lo bf16 pairs in the fp32x3 mode."""
    return PACK_LINEAR_X3 if code == F32X3 else PACK_LINEAR


class pack_scope:
    """``with ops.pack_scope(module):`` around a network forward (RDSTSR.forward does it)."""

    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        global _plan_active, _plan_recording
        self.outer = (_plan_active, _plan_recording)
        if self.outer != (None, None):     # nested network forwards share the outer scope
            return self
        plan = _PLANS.get(self.owner)
        if plan is not None and not plan.valid(self.owner):
            _PLANS.pop(self.owner, None)     # parameters moved or changed dtype: frees the stale arena and weight copies
            plan = None
        if plan is not None:
            plan.misses = 0
            plan.run()
            _plan_active = plan
        else:
            _plan_recording = PackPlan()
        return self

    def __exit__(self, *exc):
        global _plan_active, _plan_recording
        if self.outer != (None, None):
            return False
        rec, act = _plan_recording, _plan_active
        _plan_active = _plan_recording = None
        if rec is not None and exc[0] is None:
            dev = next(self.owner.parameters()).device
            if rec.finalize(dev, self.owner):
                _PLANS[self.owner] = rec
            else:
                _PLANS.pop(self.owner, None)
        if act is not None and act.misses:
            _PLANS.pop(self.owner, None)         # the forward took another path than the recorded one: rediscover
        return False


def _packed_workspace(kind, w, lw, lb, b, N, K, s, nbytes, device):
    """(workspace pointer holder, pointer, workspace_bytes) for an op that reads a packed image of (w, lw, lb, b)."""
    if _plan_active is not None:
        p = _plan_active.lookup(kind, w, lw, lb, b, N, K, s)
        if p is not None:
            return None, p, _lib.PREPACKED
    if _plan_recording is not None:
        _plan_recording.record(kind, w, lw, lb, b, N, K, s, (int(nbytes) + 255) // 256 * 256)
    wsp = _workspace(nbytes, device)
    return wsp, wsp.data_ptr(), int(nbytes)


# The weight-gradient branch and the data-gradient branch of a Linear / conv backward are independent
# and each under-fills the chip, so they run concurrently: wgrad on a side HIP stream, dgrad on the
# current one, joined before the op returns (fork/join is capturable into a HIP graph).
_side_streams: dict = {}
TWO_STREAM_BACKWARD = os.environ.get("RDST_TWO_STREAM", "0") != "0"   # env switch: profiling with clean kernel durations
MLP_FUSED = os.environ.get("RDST_MLP_FUSED", "1") != "0"   # K7 (fused Mlp kernels) on/off
ATTN_LSE = os.environ.get("RDST_ATTN_LSE", "1") != "0"       # window 16: keep the forward's row statistics for the backward (rdst_wattn_*_lse)
X3_STREAM = os.environ.get("RDST_X3_STREAM", "1") != "0"     # fp32x3: the streaming Linear kernels on prepacked hi / lo images (lin3x_mfma.hip, lnlin3x_mfma.hip) on/off
ATTN_FUSED = os.environ.get("RDST_ATTN_FUSED", "1") != "0"   # K8 (LayerNorm + qkv -> attention -> proj + shortcut in one launch) on/off


def _side_stream(device) -> "torch.cuda.Stream":
    key = (device.index if device.index is not None else torch.cuda.current_device())
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return st


# ------------------------------------------------------------------------------------------------
# K1 / K2: fused window attention
# ------------------------------------------------------------------------------------------------
class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, table, mask, H, W, heads, ws, shift, scale, attn_drop=0.0, seed=None):
        _need_gpu(qkv, table, mask)
        lib = _lib.load()
        C = qkv.shape[-1] // 3
        B = qkv.numel() // (H * W * 3 * C)
        qkv_r, ld = _rows(qkv)
        tab = _param(table)
        msk = _param(mask)
        nw = 0 if msk is None else msk.shape[0]
        out = torch.empty(qkv.shape[:-1] + (C,), dtype=qkv.dtype, device=qkv.device)
        code = _dtype_code(qkv)
        if attn_drop > 0.0:   # training-mode attention dropout: the shape-generic kernels with a counter-based mask
            _lib.check(lib.rdst_wattn_fwd_drop(qkv_r.data_ptr(), ld, tab.data_ptr(), _ptr(msk), nw, out.data_ptr(), C, B, H,
                                               W, C, heads, ws, shift, float(scale), code, float(attn_drop),
                                               seed.data_ptr(), _stream()), "rdst_wattn_fwd_drop")
        else:
            _lib.check(lib.rdst_wattn_fwd(qkv_r.data_ptr(), ld, tab.data_ptr(), _ptr(msk), nw, out.data_ptr(), C, B, H,
                                          W, C, heads, ws, shift, float(scale), code, _stream()),
                       "rdst_wattn_fwd")
        ctx.save_for_backward(qkv_r, tab, msk, seed)
        ctx.geom = (B, H, W, C, heads, ws, shift, float(scale), ld, nw, float(attn_drop), code)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, tab, msk, seed = ctx.saved_tensors
        B, H, W, C, heads, ws, shift, scale, ld, nw, attn_drop, code = ctx.geom
        lib = _lib.load()
        dout_r, ldd = _rows(dout)
        dqkv = torch.empty(qkv.shape[:-1] + (3 * C,), dtype=qkv.dtype, device=qkv.device)
        dtable = _grad_like(tab)
        nbytes = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
        wsp = _workspace(nbytes, qkv.device)
        if _ReduceBatch.depth > 0:   # an outer batch is open: the d(table) slabs are summed when it ends
            _ReduceBatch.keep.append(wsp)
        if attn_drop > 0.0:
            _lib.check(lib.rdst_wattn_bwd_drop(qkv.data_ptr(), ld, tab.data_ptr(), _ptr(msk), nw, dout_r.data_ptr(), ldd,
                                               dqkv.data_ptr(), 3 * C, dtable.data_ptr(), wsp.data_ptr(), nbytes, B, H, W,
                                               C, heads, ws, shift, scale, code, attn_drop, seed.data_ptr(),
                                               _stream()), "rdst_wattn_bwd_drop")
        else:
            _lib.check(lib.rdst_wattn_bwd(qkv.data_ptr(), ld, tab.data_ptr(), _ptr(msk), nw, dout_r.data_ptr(), ldd,
                                          dqkv.data_ptr(), 3 * C, dtable.data_ptr(), wsp.data_ptr(), nbytes, B, H, W,
                                          C, heads, ws, shift, scale, code, _stream()), "rdst_wattn_bwd")
        _ReduceBatch.settle(lib)
        return dqkv, dtable, None, None, None, None, None, None, None, None, None


def draw_seed(device) -> torch.Tensor:
    """One 64-bit seed on `device` from torch's generator (so torch.manual_seed governs it and a HIP-graph replay of the
    call draws a new one): the seed argument of the attention-dropout entry points."""
    return torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=device)


def window_attention(qkv: torch.Tensor, table: torch.Tensor, H: int, W: int, heads: int, ws: int, shift: int,
                     scale: float, mask: Optional[torch.Tensor] = None, attn_drop: float = 0.0,
                     seed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused roll + window partition + (q*scale)k^T + relative-position bias + shift mask + softmax
    + @v + window reverse + un-roll on token-major qkv (..., 3C) -> (..., C).
    Replaces networks/swin_transformer_sr.py:244-267 with :117-138 inside (minus the Linears).
    ``mask`` (nW,N,N) overrides the analytic shifted-window mask (standalone WindowAttention API).
    ``attn_drop`` > 0 (the caller passes it in training only): nn.Dropout on the softmax output (:136) with a counter-based
    mask keyed by ``seed`` (a 1-element int64 tensor on the device; drawn from torch's generator when not given)."""
    if attn_drop and attn_drop > 0.0:
        if not (0.0 < attn_drop < 1.0):
            raise ValueError(f"window_attention: attn_drop={attn_drop} must be in [0, 1)")
        if seed is None:
            seed = draw_seed(qkv.device)
        return _WindowAttention.apply(qkv, table, mask, H, W, heads, ws, shift, scale, float(attn_drop), seed)
    return _WindowAttention.apply(qkv, table, mask, H, W, heads, ws, shift, scale)


def _linear_bwd_call(lib, x, ldx, lw, lb, stats, in_act, w, dy, lddy, dx, lddx, dx_add, ld_add, dw, db, dlw, dlb,
                     M, K, N, out_scale, code, dev, join=True, keep=None, dx_add2=None, ld_add2=0):
    """rdst_ln_linear_bwd with the weight-gradient half on the side stream (see TWO_STREAM_BACKWARD).
    join=False leaves the side stream un-joined (the caller joins once, later) and parks the workspace in
    `keep` so it outlives the asynchronous kernel.  dx_add2: a second (strided) addend of dx — folded into the
    kernel where it takes one (rdst_ln_linear_bwd2), added afterwards otherwise."""
    nbytes = lib.rdst_ln_linear_bwd_workspace(M, K, N)

    def call(dx_, add_, dw_, db_, dlw_, dlb_, wsp_):
        if dx_add2 is not None and dx_ is not None:
            rc = lib.rdst_ln_linear_bwd2(x.data_ptr(), ldx, _ptr(lw), _ptr(lb), _ptr(stats), in_act, _ptr(w),
                                         dy.data_ptr(), lddy, _ptr(dx_), lddx, _ptr(add_), ld_add, _ptr(dw_), _ptr(db_),
                                         _ptr(dlw_), _ptr(dlb_), wsp_.data_ptr(), nbytes, M, K, N, out_scale, code,
                                         _stream(), dx_add2.data_ptr(), ld_add2)
            if rc != _lib.ENOTSUP:
                _lib.check(rc, "rdst_ln_linear_bwd2")
                return
        _lib.check(lib.rdst_ln_linear_bwd(x.data_ptr(), ldx, _ptr(lw), _ptr(lb), _ptr(stats), in_act, _ptr(w),
                                          dy.data_ptr(), lddy, _ptr(dx_), lddx, _ptr(add_), ld_add, _ptr(dw_), _ptr(db_),
                                          _ptr(dlw_), _ptr(dlb_), wsp_.data_ptr(), nbytes, M, K, N, out_scale, code,
                                          _stream()), "rdst_ln_linear_bwd")
        if dx_add2 is not None and dx_ is not None:
            dx_.add_(dx_add2)

    wgrad = dw is not None or db is not None
    dgrad = dx is not None or dlw is not None or dlb is not None
    if TWO_STREAM_BACKWARD and wgrad and dgrad:
        cur, side = torch.cuda.current_stream(), _side_stream(dev)
        wsp_w, wsp_d = _workspace(nbytes, dev), _workspace(nbytes, dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            call(None, None, dw, db, None, None, wsp_w)
        call(dx, dx_add, None, None, dlw, dlb, wsp_d)
        if join:
            cur.wait_stream(side)
        elif keep is not None:
            keep.append(wsp_w)
    elif wgrad or dgrad:
        wsp = _workspace(nbytes, dev)
        if keep is not None:   # a reduction batch is open: the slabs are read when it ends
            keep.append(wsp)
        call(dx, dx_add, dw, db, dlw, dlb, wsp)


# ------------------------------------------------------------------------------------------------
# K3: (LayerNorm | activation ->) Linear (-> *scale + residual)
# ------------------------------------------------------------------------------------------------
class _LnLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, weight, bias, residual, in_act, out_scale, out_slot=None):
        _need_gpu(x, ln_w, ln_b, weight, bias, residual)
        lib = _lib.load()
        K = x.shape[-1]
        N = K if weight is None else weight.shape[0]
        M = x.numel() // K
        x_r, ldx = _rows(x)
        lw, lb, w, b = _param(ln_w), _param(ln_b), _param(weight), _param(bias)
        if out_slot is None:
            y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
            ldy = N
        else:   # (DenseBuffer, first channel): the kernel writes its N channels straight into the dense buffer
            y = out_slot[0].slot(out_slot[1], N)
            ldy = out_slot[0].width
            if y.shape[:-1] != x.shape[:-1] or y.dtype != x.dtype:
                raise ValueError("rdst_amd.ln_linear: out_slot does not match the input's rows / dtype")
        r_r, ldr = (None, 0)
        if residual is not None:
            if residual.dtype != x.dtype:
                raise TypeError("rdst_amd.ln_linear: residual dtype differs from the activation dtype")
            r_r, ldr = _rows(residual)
        stats = torch.empty((M, 2), dtype=torch.float32, device=x.device) if lw is not None else None
        code = _dtype_code(x)
        if w is not None and (X3_STREAM or code != F32X3) and lib.rdst_ln_linear_fwd_packable(
                K, N, int(lw is not None), int(r_r is not None), int(in_act), code):
            _wsp, wptr, nws = _packed_workspace(_linear_pack_kind(code), w, lw, lb, b, N, K, out_scale,
                                                lib.rdst_ln_linear_fwd_workspace2(K, N, code), x.device)
        else:
            _wsp, wptr, nws = None, None, 0
        _lib.check(lib.rdst_ln_linear_fwd(x_r.data_ptr(), ldx, _ptr(lw), _ptr(lb), int(in_act), _ptr(w), _ptr(b),
                                          _ptr(r_r), ldr, y.data_ptr(), ldy, _ptr(stats), wptr, nws, M, K, N,
                                          float(out_scale), code, _stream()), "rdst_ln_linear_fwd")
        ctx.save_for_backward(x_r, lw, lb, w, stats)
        ctx.bias_ref = b   # only its address is used in backward (destination lookup of d(bias))
        ctx.meta = (M, K, N, ldx, int(in_act), float(out_scale), bias is not None, residual is not None, code)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, lw, lb, w, stats = ctx.saved_tensors
        M, K, N, ldx, in_act, out_scale, has_bias, has_res, code = ctx.meta
        lib = _lib.load()
        dy_r, lddy = _rows(dy)
        need = ctx.needs_input_grad
        dev = x.device
        dx = torch.empty(x.shape[:-1] + (K,), dtype=x.dtype, device=dev) if need[0] else None
        dlw = _grad_like(lw) if (lw is not None and need[1]) else None
        dlb = _grad_like(lb) if (lb is not None and need[2]) else None
        dw = _grad_like(w) if (w is not None and need[3]) else None
        db = (_grad_like(ctx.bias_ref) if ctx.bias_ref is not None
              else _fresh_grad(torch.empty(N, dtype=torch.float32, device=dev))) if (has_bias and need[4]) else None
        _linear_bwd_call(lib, x, ldx, lw, lb, stats, in_act, w, dy_r, lddy, dx, K, None, 0, dw, db, dlw, dlb, M, K, N,
                         out_scale, code, dev, keep=_ReduceBatch.keep if _ReduceBatch.depth > 0 else None)
        _ReduceBatch.settle(lib)
        dres = dy if (has_res and need[5]) else None
        return dx, dlw, dlb, dw, db, dres, None, None, None


class DenseBuffer:
    """The dense concat buffer of an RDSTB (rdst_variations.py:339-340, :436-441): ONE (rows, width) tensor whose
    channel ranges the DenseSTLayers fill in place, so no torch.cat ever copies the growing prefix.  It is not an
    autograd tensor; the ops write it through raw pointers (no version-counter traffic) and hand out views."""

    def __init__(self, lead, width, dtype, device):
        self.width = int(width)
        self.t = torch.empty(tuple(lead) + (self.width,), dtype=dtype, device=device)

    def slot(self, c0, n):
        return self.t[..., c0:c0 + n]

    def fill(self, c0, src):
        """Copy values into channels [c0, c0 + C) WITHOUT touching the buffer's autograd version counter (``.data``): views of
        the buffer that custom Functions returned earlier (the patch norm's / a fusion conv's out_slot) stay usable — a plain
        ``slot().copy_()`` marks their base as modified in place and autograd then refuses them."""
        self.t.data[..., c0:c0 + src.shape[-1]].copy_(src.detach())


class _IntoDense(torch.autograd.Function):
    """x -> the first channels of a dense buffer (the only copy an RDSTB makes)."""

    @staticmethod
    def forward(ctx, x, buf):
        v = buf.slot(0, x.shape[-1])
        v.copy_(x)
        return v


    @staticmethod
    def backward(ctx, g):
        return g, None


class _ReduceBatchState(type):
    """``_ReduceBatch.depth / .keep / .fresh`` live PER THREAD, like the C side's queues (csrc/reduce_batch.hip keeps
    them ``thread_local``): autograd runs each device's backward on its own thread, so two devices in one process (or two
    backward passes on two Python threads) never share a counter while their C queues are separate."""
    _tls = threading.local()

    def _state(cls):
        st = cls._tls.__dict__
        if "depth" not in st:
            st.update(depth=0, keep=[], fresh=0, owner=-1, mixed=False, epoch=cls.EPOCH)
        return st

    # bumped by reset_backward_state(): a thread whose open batch was begun in an older epoch drops it (abort) before it
    # queues, flushes or nests anything - how the main thread reaches the autograd device thread's batch after a failed pass
    EPOCH = 0

    depth = property(lambda cls: cls._state()["depth"], lambda cls, v: cls._state().__setitem__("depth", v))
    keep = property(lambda cls: cls._state()["keep"], lambda cls, v: cls._state().__setitem__("keep", v))
    fresh = property(lambda cls: cls._state()["fresh"], lambda cls, v: cls._state().__setitem__("fresh", v))
    # the autograd graph task that opened the batch, and whether a node of ANOTHER task ran while it was open: the engine's
    # device thread serves every concurrent backward() of that device, so nodes of two passes can interleave
    owner = property(lambda cls: cls._state()["owner"], lambda cls, v: cls._state().__setitem__("owner", v))
    mixed = property(lambda cls: cls._state()["mixed"], lambda cls, v: cls._state().__setitem__("mixed", v))
    epoch = property(lambda cls: cls._state()["epoch"], lambda cls, v: cls._state().__setitem__("epoch", v))


class _ReduceBatch(metaclass=_ReduceBatchState):
    """Nesting-aware rdst_reduce_batch_begin / _end.  A Swin block's backward opens a batch for its own four ops; a
    DenseSTLayer (dense join with a GradSink) opens an OUTER one in the join's backward — the first node of the layer's
    backward — which the layer's first Swin block closes at the end of its own — the last node — so the slab sums and
    LayerNorm finishes of the whole layer (two blocks + the tail Linear) run as 2 launches instead of 6.  `keep` holds
    the ops' workspaces (the slabs) until the batch has run.

    A batch that spans autograd nodes defers WRITES of parameter gradients past the node that returns them.  That is
    only sound for destinations nobody reads before the batch ends: the flat-bucket views a detach_grads() bracket
    offers (rdst_amd.dp: p.grad is None, AccumulateGrad keeps the tensor without looking at it).  Any other destination
    (`fresh`: p.grad already defined -> autograd runs ``p.grad += g`` right after the node — bucket.zero() + backward,
    gradient accumulation, zero_grad(set_to_none=False) —, or a parameter used twice) makes the node `settle` before
    it returns: the queued reductions run at once and the outer batch goes on empty."""

    @staticmethod
    def _task():
        return torch._C._current_graph_task_id()

    @staticmethod
    def _foreign():
        """A node of another backward pass runs inside this thread's open batch (two Python threads called backward() on
        the same device at once): from here until the batch closes every node flushes what is queued before it returns,
        so no pass ever returns with reductions of its own still parked in the other pass's batch."""
        if _ReduceBatch.depth > 0 and _ReduceBatch._task() != _ReduceBatch.owner:
            _ReduceBatch.mixed = True
        return _ReduceBatch.mixed

    @staticmethod
    def _drop_stale(lib):
        """This thread's batch was opened before the last reset_backward_state(): the pass it belonged to is dead."""
        if _ReduceBatch.epoch != type(_ReduceBatch).EPOCH:
            _ReduceBatch.abandon(lib)
            _ReduceBatch.epoch = type(_ReduceBatch).EPOCH

    @staticmethod
    def begin(lib):
        _ReduceBatch._drop_stale(lib)
        if _ReduceBatch.depth == 0:
            _lib.check(lib.rdst_reduce_batch_begin(), "rdst_reduce_batch_begin")
            _ReduceBatch.owner = _ReduceBatch._task()
            _ReduceBatch.mixed = False
        _ReduceBatch.depth += 1

    @staticmethod
    def begin_layer(lib):
        """The dense join's outer batch (first node of a DenseSTLayer's backward).  Inside ONE pass no batch is open here,
        so an open one is either a leftover of this pass's task (a node raised and the pass was re-entered: drop it) or the
        batch of ANOTHER live pass served by the same device thread: that one keeps its queued reductions - this pass
        nests inside it and from here on every node flushes before it returns (`mixed`)."""
        _ReduceBatch._drop_stale(lib)
        if _ReduceBatch.depth > 0:
            if _ReduceBatch.owner == _ReduceBatch._task():
                _ReduceBatch.abandon(lib)
            else:
                _ReduceBatch.mixed = True
        _ReduceBatch.begin(lib)

    @staticmethod
    def end(lib, keep=None):
        if keep:
            _ReduceBatch.keep.extend(keep)
        _ReduceBatch.depth -= 1
        if _ReduceBatch.depth == 0:
            try:
                _lib.check(lib.rdst_reduce_batch_end(_stream()), "rdst_reduce_batch_end")
            finally:
                _ReduceBatch.keep = []
                _ReduceBatch.fresh = 0

    @staticmethod
    def settle(lib):
        """Last statement of a node's backward: if a batch is still open around this node and the node handed out a
        gradient destination that autograd may read on return, run what is queued now (and keep the batch open)."""
        _ReduceBatch._drop_stale(lib)
        if _ReduceBatch.depth > 0 and (_ReduceBatch.fresh or _ReduceBatch._foreign()):
            try:
                _lib.check(lib.rdst_reduce_batch_end(_stream()), "rdst_reduce_batch_end")
            finally:
                _ReduceBatch.keep = []
                _ReduceBatch.fresh = 0
            _lib.check(lib.rdst_reduce_batch_begin(), "rdst_reduce_batch_begin")
        elif _ReduceBatch.depth == 0:
            _ReduceBatch.fresh = 0

    @staticmethod
    def abandon(lib):
        """Forget whatever a failed / partial backward left open: its queued reductions are DROPPED — the slabs and the
        outputs they name belonged to that backward and may be freed (rdst_reduce_batch_abort)."""
        if _ReduceBatch.depth > 0:
            _lib.check(lib.rdst_reduce_batch_abort(), "rdst_reduce_batch_abort")
        _ReduceBatch.depth = 0
        _ReduceBatch.keep = []
        _ReduceBatch.fresh = 0
        _ReduceBatch.mixed = False


def reset_backward_state() -> None:
    """After a backward pass that did not run to its end (an exception inside a node, a failed HIP-graph capture): every
    thread drops the reduction batch it still holds open - its queued jobs name workspaces of the dead pass - the next
    time it touches one.  The calling thread's is dropped at once."""
    type(_ReduceBatch).EPOCH += 1
    if _lib.loaded():
        _ReduceBatch._drop_stale(_lib.load())


class GradSink:
    """Carries the prefix slice of a dense join's gradient to the Swin block that consumed the prefix: `prefix` feeds
    the DenseSTLayer's body AND the join, so autograd would add the body's dX and this strided slice in a separate
    kernel (24 per E1 step).  With a sink the join returns None for the prefix and the body's first block adds the
    slice inside its last backward kernel.  The join's backward always runs first (it consumes what the body made)."""

    def __init__(self):
        self.extra = None
        self.batch_open = False   # the join opened the layer's reduction batch (_ReduceBatch); the taker closes it

    def take(self):
        e, self.extra = self.extra, None
        return e

    def take_batch(self):
        b, self.batch_open = self.batch_open, False
        return b


class _DenseJoin(torch.autograd.Function):
    """cat(prefix, new) where both already lie side by side in the dense buffer: returns the wider view."""

    @staticmethod
    def forward(ctx, prefix, new, buf, sink=None):
        ctx.c = prefix.shape[-1]
        ctx.sink = sink
        return buf.slot(0, prefix.shape[-1] + new.shape[-1])

    @staticmethod
    def backward(ctx, g):
        if ctx.sink is not None and ctx.needs_input_grad[0]:
            ctx.sink.extra = g[..., :ctx.c]
            if g.is_cuda and not TWO_STREAM_BACKWARD:
                lib = _lib.load()
                _ReduceBatch.begin_layer(lib)
                ctx.sink.batch_open = True
            return None, g[..., ctx.c:], None, None
        return g[..., :ctx.c], g[..., ctx.c:], None, None


def into_dense(x: torch.Tensor, buf: DenseBuffer) -> torch.Tensor:
    if x.data_ptr() == buf.t.data_ptr() and x.stride()[:-1] == buf.t.stride()[:-1] and x.shape[:-1] == buf.t.shape[:-1]:
        return x   # its producer already wrote it into the buffer's first channels (out_slot)
    return _IntoDense.apply(x, buf)


def dense_join(prefix: torch.Tensor, new: torch.Tensor, buf: DenseBuffer, sink: "GradSink" = None) -> torch.Tensor:
    return _DenseJoin.apply(prefix, new, buf, sink)


def ln_linear(x: torch.Tensor, ln_w: Optional[torch.Tensor], ln_b: Optional[torch.Tensor],
              weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], *, in_act: int = ACT_NONE,
              residual: Optional[torch.Tensor] = None, out_scale: float = 1.0, out_slot=None) -> torch.Tensor:
    """y = (f(x) @ weight^T + bias) * out_scale + residual, f = LayerNorm (ln_w given) or the
    activation ``in_act`` or identity; weight None = LayerNorm only.  One fused HIP op replacing the
    reference's LayerNorm/Linear/GELU/add sequences (see include/rdst_hip.h, K3)."""
    return _LnLinear.apply(x, ln_w, ln_b, weight, bias, residual, in_act, out_scale, out_slot)


# ------------------------------------------------------------------------------------------------
# A whole Swin block as ONE autograd node (5 forward kernels, backward chain called directly)
# ------------------------------------------------------------------------------------------------
class _SwinBlock(torch.autograd.Function):
    """y = x1 + fc2(GELU(fc1(LN2(x1)))),  x1 = x + proj(WindowAttention(qkv(LN1(x)))).
    One node instead of five: the backward runs the kernels in dependency order and folds the two
    residual fan-out sums into the dgrad kernels (dX = dX_add + ...), so no gradient-accumulation add
    kernels and no intermediate gradient copies are launched."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qkvb, table, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, H, W, heads, ws,
                shift, scale, sink=None):
        ctx.sink = sink
        _need_gpu(x, qkvw, table, projw, fc1w, fc2w)
        lib = _lib.load()
        C = x.shape[-1]
        M = x.numel() // C
        B = M // (H * W)
        hid = fc1w.shape[0]
        code = _dtype_code(x)
        dev, dt = x.device, x.dtype
        x_r, ldx = _rows(x)
        P = [_param(t) for t in (n1w, n1b, qkvw, qkvb, table, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b)]
        n1w_, n1b_, qkvw_, qkvb_, tab_, projw_, projb_, n2w_, n2b_, fc1w_, fc1b_, fc2w_, fc2b_ = P
        lead = x.shape[:-1]
        st = _stream()

        def lin(xp, ld, lw, lb, act, w, b, rp, ldr, out, N, stats, K):
            if (X3_STREAM or code != F32X3) and lib.rdst_ln_linear_fwd_packable(K, N, int(lw is not None), int(rp is not None), act, code):
                _wsp, wptr, nws = _packed_workspace(_linear_pack_kind(code), w, lw, lb, b, N, K, 1.0,
                                                    lib.rdst_ln_linear_fwd_workspace2(K, N, code), dev)
            else:
                _wsp, wptr, nws = None, None, 0
            _lib.check(lib.rdst_ln_linear_fwd(xp, ld, _ptr(lw), _ptr(lb), act, w.data_ptr(), _ptr(b), rp, ldr,
                                              out.data_ptr(), N, _ptr(stats), wptr, nws, M, K, N, 1.0, code, st),
                       "rdst_ln_linear_fwd")

        stats1 = torch.empty((M, 2), dtype=torch.float32, device=dev) if n1w_ is not None else None
        qkv = torch.empty(lead + (3 * C,), dtype=dt, device=dev)
        a = torch.empty(lead + (C,), dtype=dt, device=dev)
        x1 = torch.empty(lead + (C,), dtype=dt, device=dev)
        fused_attn = (ATTN_FUSED and n1w_ is not None and n1b_ is not None
                      and bool(lib.rdst_swin_attn_fwd_supported(C, heads, ws, code)))
        if fused_attn:
            # K8: norm1 + qkv -> window attention -> proj + shortcut as ONE launch (qkv, a, x1, stats1 for the backward)
            nbytes = lib.rdst_swin_attn_fwd_workspace(C)
            wptr, nws, _wsp = None, 0, None
            if _plan_active is not None:
                wptr = _plan_active.lookup_swinattn(qkvw_, n1w_, n1b_, qkvb_, projw_, projb_)
                nws = _lib.PREPACKED
            if wptr is None:
                nbp = lib.rdst_ln_linear_fwd_workspace(C, C)
                if _plan_recording is not None:
                    _plan_recording.record_swinattn(qkvw_, n1w_, n1b_, qkvb_, projw_, projb_, C, nbytes - nbp, nbp)
                _wsp = _workspace(nbytes, dev)
                wptr, nws = _wsp.data_ptr(), nbytes
            rc = lib.rdst_swin_attn_fwd(x_r.data_ptr(), ldx, n1w_.data_ptr(), n1b_.data_ptr(), qkvw_.data_ptr(), _ptr(qkvb_),
                                        tab_.data_ptr(), projw_.data_ptr(), _ptr(projb_), qkv.data_ptr(), 3 * C, a.data_ptr(), C,
                                        x1.data_ptr(), C, stats1.data_ptr(), wptr, nws, B, H, W, C, heads, ws, shift,
                                        float(scale), code, st)
            if rc == _lib.ENOTSUP:
                fused_attn = False
            else:
                _lib.check(rc, "rdst_swin_attn_fwd")
        nlse = None
        if not fused_attn:
            lin(x_r.data_ptr(), ldx, n1w_, n1b_, ACT_NONE, qkvw_, qkvb_, None, 0, qkv, 3 * C, stats1, C)
            if ATTN_LSE and ws == 16 and code == BF16:
                # window 16: keep the row statistics, the backward's first pass then streams its key tiles (rdst_wattn_bwd_lse)
                nlse = torch.empty((M, heads), dtype=torch.float32, device=dev)
                rc = lib.rdst_wattn_fwd_lse(qkv.data_ptr(), 3 * C, tab_.data_ptr(), a.data_ptr(), C, nlse.data_ptr(), B, H, W, C,
                                            heads, ws, shift, float(scale), code, _stream())
                if rc == _lib.ENOTSUP:
                    nlse = None
                else:
                    _lib.check(rc, "rdst_wattn_fwd_lse")
            if nlse is None:
                _lib.check(lib.rdst_wattn_fwd(qkv.data_ptr(), 3 * C, tab_.data_ptr(), None, 0, a.data_ptr(), C, B, H, W, C,
                                              heads, ws, shift, float(scale), code, _stream()), "rdst_wattn_fwd")
            lin(a.data_ptr(), C, None, None, ACT_NONE, projw_, projb_, x_r.data_ptr(), ldx, x1, C, None, C)
        stats2 = torch.empty((M, 2), dtype=torch.float32, device=dev) if n2w_ is not None else None
        y = torch.empty(lead + (C,), dtype=dt, device=dev)
        h = None
        fused_mlp = MLP_FUSED and n2w_ is not None and bool(lib.rdst_mlp_fused_supported(C, hid, code))
        if fused_mlp:
            # K7: the whole Mlp half in one kernel; the hidden activations are not kept (the backward recomputes them)
            wptr, nws, _wsp = None, 0, None
            if lib.rdst_mlp_fwd_packable(C, hid, code):
                nb1, nb2 = lib.rdst_ln_linear_fwd_workspace(C, hid), lib.rdst_ln_linear_fwd_workspace(hid, C)
                if _plan_active is not None:
                    wptr = _plan_active.lookup_mlp(fc1w_, n2w_, n2b_, fc1b_, fc2w_, fc2b_)
                    nws = _lib.PREPACKED
                if wptr is None:
                    if _plan_recording is not None:
                        _plan_recording.record_mlp(fc1w_, n2w_, n2b_, fc1b_, fc2w_, fc2b_, C, hid, nb1, nb2)
                    nws = lib.rdst_mlp_fwd_workspace(C, hid)
                    _wsp = _workspace(nws, dev)
                    wptr = _wsp.data_ptr()
            rc = lib.rdst_mlp_fwd(x1.data_ptr(), C, n2w_.data_ptr(), n2b_.data_ptr(), fc1w_.data_ptr(), _ptr(fc1b_),
                                  fc2w_.data_ptr(), _ptr(fc2b_), y.data_ptr(), C, stats2.data_ptr(), wptr, nws, M, C, hid, code, st)
            if rc == _lib.ENOTSUP:
                fused_mlp = False
            else:
                _lib.check(rc, "rdst_mlp_fwd")
        if not fused_mlp:
            h = torch.empty(lead + (hid,), dtype=dt, device=dev)
            lin(x1.data_ptr(), C, n2w_, n2b_, ACT_NONE, fc1w_, fc1b_, None, 0, h, hid, stats2, C)
            lin(h.data_ptr(), hid, None, None, ACT_GELU, fc2w_, fc2b_, x1.data_ptr(), C, y, C, None, hid)
        ctx.save_for_backward(x_r, stats1, qkv, a, x1, stats2, h, nlse, *P)
        ctx.meta = (M, B, H, W, C, hid, heads, ws, shift, float(scale), ldx, code)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, stats1, qkv, a, x1, stats2, h, nlse, n1w, n1b, qkvw, qkvb, tab, projw, projb, n2w, n2b, fc1w, fc1b, fc2w,
         fc2b) = ctx.saved_tensors
        M, B, H, W, C, hid, heads, ws, shift, scale, ldx, code = ctx.meta
        lib = _lib.load()
        dev, dt = x.device, x.dtype
        need = ctx.needs_input_grad
        dy_r, lddy = _rows(dy)

        def g(t, flag):
            return _grad_like(t) if (t is not None and flag) else None

        dn1w, dn1b, dqkvw, dqkvb = g(n1w, need[1]), g(n1b, need[2]), g(qkvw, need[3]), g(qkvb, need[4])
        dprojw, dprojb = g(projw, need[6]), g(projb, need[7])
        dn2w, dn2b, dfc1w, dfc1b = g(n2w, need[8]), g(n2b, need[9]), g(fc1w, need[10]), g(fc1b, need[11])
        dfc2w, dfc2b = g(fc2w, need[12]), g(fc2b, need[13])
        # The four weight-gradient kernels are leaves: they go to the side stream and are joined ONCE at the
        # end of this block's backward, so they overlap the whole data-gradient chain below.  Everything they
        # read (saved activations, dy, dh, dx1, dqkv, their workspaces) stays referenced until then.
        keep = []
        # the slab reductions of the four ops below are recorded and run as two launches at the end (rdst_reduce_batch_*);
        # their workspaces are locals of this function, alive until then
        batched = not TWO_STREAM_BACKWARD
        outer = batched and ctx.sink is not None and ctx.sink.take_batch()   # this block ends its DenseSTLayer's batch too
        if batched:
            _ReduceBatch.begin(lib)
        try:
            return _SwinBlock._backward_body(ctx, lib, dy_r, lddy, need, keep, dn1w, dn1b, dqkvw, dqkvb, dprojw, dprojb, dn2w,
                                             dn2b, dfc1w, dfc1b, dfc2w, dfc2b)
        finally:
            if batched:
                _ReduceBatch.end(lib, keep)
            if outer:
                _ReduceBatch.end(lib)
            _ReduceBatch.settle(lib)   # (a later block of a DenseSTLayer: the layer's batch is still open around it)

    @staticmethod
    def _backward_body(ctx, lib, dy_r, lddy, need, keep, dn1w, dn1b, dqkvw, dqkvb, dprojw, dprojb, dn2w, dn2b, dfc1w, dfc1b,
                       dfc2w, dfc2b):
        (x, stats1, qkv, a, x1, stats2, h, nlse, n1w, n1b, qkvw, qkvb, tab, projw, projb, n2w, n2b, fc1w, fc1b, fc2w,
         fc2b) = ctx.saved_tensors
        M, B, H, W, C, hid, heads, ws, shift, scale, ldx, code = ctx.meta
        dev, dt = x.device, x.dtype
        dx1 = torch.empty_like(x1)
        # K7 when the forward was fused (h was never written) or whenever every Mlp gradient is wanted anyway
        fused_mlp = h is None or (MLP_FUSED and all(t is not None for t in (dn2w, dn2b, dfc1w, dfc1b, dfc2w, dfc2b))
                                  and bool(lib.rdst_mlp_fused_supported(C, hid, code)))
        if fused_mlp:
            # the whole Mlp backward in one pass over (x1, dy); the hidden activations are recomputed.  The kernel
            # writes every parameter gradient: the ones nobody asked for land in scratch.
            def out(t, like, n=None):
                if t is not None:
                    return t
                t = torch.empty_like(like) if like is not None else torch.empty(n, dtype=torch.float32, device=dev)
                keep.append(t)   # scratch destination of a deferred reduction: alive until the batch has run
                return t
            o = [out(dfc1w, fc1w), out(dfc1b, fc1b, hid), out(dfc2w, fc2w), out(dfc2b, fc2b, C), out(dn2w, n2w),
                 out(dn2b, n2b)]
            nb = lib.rdst_mlp_bwd_workspace(M, C, hid)
            wsp_m = _workspace(nb, dev)
            keep.append(wsp_m)   # read by the batched reductions at the end of backward()
            rc = lib.rdst_mlp_bwd(x1.data_ptr(), C, n2w.data_ptr(), n2b.data_ptr(), stats2.data_ptr(), fc1w.data_ptr(),
                                  _ptr(fc1b), fc2w.data_ptr(), dy_r.data_ptr(), lddy, dx1.data_ptr(), C,
                                  o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(),
                                  o[5].data_ptr(), wsp_m.data_ptr(), nb, M, C, hid, code, _stream())
            if rc == _lib.ENOTSUP and h is None:
                # The forward ran fused (h was never written), so there is no composed path to fall back to; what the
                # kernel can refuse at this point is the ALIGNMENT of dy (a strided gradient slice with an odd channel
                # offset): hand it an aligned contiguous copy (ld = C) — same kernel, same result.
                dy_c = dy_r.clone(memory_format=torch.contiguous_format)
                keep.append(dy_c)
                rc = lib.rdst_mlp_bwd(x1.data_ptr(), C, n2w.data_ptr(), n2b.data_ptr(), stats2.data_ptr(), fc1w.data_ptr(),
                                      _ptr(fc1b), fc2w.data_ptr(), dy_c.data_ptr(), C, dx1.data_ptr(), C,
                                      o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(),
                                      o[5].data_ptr(), wsp_m.data_ptr(), nb, M, C, hid, code, _stream())
            if rc == _lib.ENOTSUP and h is not None:
                fused_mlp = False
            else:
                _lib.check(rc, "rdst_mlp_bwd")
        if not fused_mlp:
            # fc2 (reads h through GELU):  dh = (dy W2) * gelu'(h)
            dh = torch.empty_like(h)
            _linear_bwd_call(lib, h, hid, None, None, None, ACT_GELU, fc2w, dy_r, lddy, dh, hid, None, 0, dfc2w, dfc2b,
                             None, None, M, hid, C, 1.0, code, dev, join=False, keep=keep)
            # LN2 + fc1, plus the residual fan-out of x1:  dx1 = dy + LN2'(dh W1)
            _linear_bwd_call(lib, x1, C, n2w, n2b, stats2, ACT_NONE, fc1w, dh, hid, dx1, C, dy_r, lddy, dfc1w, dfc1b,
                             dn2w, dn2b, M, C, hid, 1.0, code, dev, join=False, keep=keep)
        # proj:  da = dx1 Wp
        da = torch.empty_like(a)
        _linear_bwd_call(lib, a, C, None, None, None, ACT_NONE, projw, dx1, C, da, C, None, 0, dprojw, dprojb, None, None,
                         M, C, C, 1.0, code, dev, join=False, keep=keep)
        # window attention
        dqkv = torch.empty_like(qkv)
        dtab = _grad_like(tab)
        nbytes = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
        wsp = _workspace(nbytes, dev)
        keep.append(wsp)
        rc = _lib.ENOTSUP
        if nlse is not None:   # window 16 with the forward's row statistics: the streaming first pass
            rc = lib.rdst_wattn_bwd_lse(qkv.data_ptr(), 3 * C, tab.data_ptr(), da.data_ptr(), C, a.data_ptr(), C, nlse.data_ptr(),
                                        dqkv.data_ptr(), 3 * C, dtab.data_ptr(), wsp.data_ptr(), nbytes, B, H, W, C, heads, ws,
                                        shift, scale, code, _stream())
            if rc != _lib.ENOTSUP:
                _lib.check(rc, "rdst_wattn_bwd_lse")
        if rc == _lib.ENOTSUP:
            _lib.check(lib.rdst_wattn_bwd(qkv.data_ptr(), 3 * C, tab.data_ptr(), None, 0, da.data_ptr(), C,
                                          dqkv.data_ptr(), 3 * C, dtab.data_ptr(), wsp.data_ptr(), nbytes, B, H, W, C,
                                          heads, ws, shift, scale, code, _stream()), "rdst_wattn_bwd")
        # LN1 + qkv, plus the residual fan-out of x:  dx = dx1 + LN1'(dqkv Wqkv)
        dx = torch.empty(x.shape, dtype=dt, device=dev) if need[0] else None
        extra = ctx.sink.take() if ctx.sink is not None else None   # the dense join's gradient slice for x (GradSink)
        if extra is not None and need[0]:
            extra, ld_extra = _rows(extra)
            keep.append(extra)
        else:
            extra, ld_extra = None, 0
        _linear_bwd_call(lib, x, ldx, n1w, n1b, stats1, ACT_NONE, qkvw, dqkv, 3 * C, dx, C, dx1 if need[0] else None, C,
                         dqkvw, dqkvb, dn1w, dn1b, M, C, 3 * C, 1.0, code, dev, join=False, keep=keep, dx_add2=extra,
                         ld_add2=ld_extra)
        if TWO_STREAM_BACKWARD:
            torch.cuda.current_stream().wait_stream(_side_stream(dev))   # the one join of this block
        if not need[5]:
            dtab = None
        return (dx, dn1w, dn1b, dqkvw, dqkvb, dtab, dprojw, dprojb, dn2w, dn2b, dfc1w, dfc1b, dfc2w, dfc2b,
                None, None, None, None, None, None, None)


def swin_block(x, n1w, n1b, qkvw, qkvb, table, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, H, W, heads, ws, shift,
               scale, sink=None):
    """SwinTransformerBlock.forward (networks/swin_transformer_sr.py:234-274) as one autograd node.  sink: a GradSink
    whose slice (set by a dense join of x) is added to dx inside the block's last backward kernel."""
    return _SwinBlock.apply(x, n1w, n1b, qkvw, qkvb, table, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, H, W, heads,
                            ws, shift, scale, sink)


# ------------------------------------------------------------------------------------------------
# K4/K5/K6: conv on rows
# ------------------------------------------------------------------------------------------------
class _ConvRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, in_act, out_scale, shuffle, out_slot=None):
        _need_gpu(x, weight, bias, residual)
        lib = _lib.load()
        if x.dim() != 4:
            raise ValueError("rdst_amd.conv_rows: x must be (B, H, W, Cin) token-major")
        B, H, W, Cin = x.shape
        Cout, cin_w, k, k2 = weight.shape
        if cin_w != Cin or k != k2:
            raise ValueError(f"rdst_amd.conv_rows: weight {tuple(weight.shape)} does not fit Cin={Cin}")
        r = int(shuffle)
        x_r, ldx = _rows(x)
        w, b = _param(weight), _param(bias)
        cy = Cout // (r * r)
        if out_slot is None:
            y, ldy = torch.empty((B, H * r, W * r, cy), dtype=x.dtype, device=x.device), cy
        else:   # (DenseBuffer, first channel): the kernel writes its channels straight into the next RDSTB's dense buffer
            y = out_slot[0].slot(out_slot[1], cy).view(B, H * r, W * r, cy)
            ldy = out_slot[0].width
        r_r, ldr = (None, 0)
        if residual is not None:
            if residual.dtype != x.dtype or tuple(residual.shape) != tuple(y.shape):
                raise ValueError("rdst_amd.conv_rows: residual must match the output shape/dtype")
            r_r, ldr = _rows(residual)
        code = _dtype_code(x)
        if (X3_STREAM or code != F32X3) and lib.rdst_conv_fwd_packable(Cin, Cout, k, r, int(r_r is not None), int(in_act), code):
            _wsp, wptr, nws = _packed_workspace(PACK_CONV3_FWD_X3 if code == F32X3 else PACK_CONV3_FWD, w, None, None, None, Cout, Cin,
                                                out_scale, lib.rdst_conv_fwd_workspace2(Cin, Cout, k, code), x.device)
        else:
            _wsp, wptr, nws = None, None, 0
        _lib.check(lib.rdst_conv_fwd(x_r.data_ptr(), ldx, int(in_act), w.data_ptr(), _ptr(b), _ptr(r_r), ldr,
                                     y.data_ptr(), ldy, wptr, nws, B, H, W, Cin, Cout, k, float(out_scale), r,
                                     code, _stream()), "rdst_conv_fwd")
        ctx.bias_ref = b   # only its address is used in backward (destination lookup of d(bias))
        ctx.save_for_backward(x_r, w)
        ctx.meta = (B, H, W, Cin, Cout, k, ldx, int(in_act), float(out_scale), r, bias is not None,
                    residual is not None, code)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, H, W, Cin, Cout, k, ldx, in_act, out_scale, r, has_bias, has_res, code = ctx.meta
        lib = _lib.load()
        dy_r, lddy = _rows(dy)
        need = ctx.needs_input_grad
        dev = x.device
        dx = torch.empty((B, H, W, Cin), dtype=x.dtype, device=dev) if need[0] else None
        dw = _grad_like(w) if need[1] else None
        db = (_grad_like(ctx.bias_ref) if ctx.bias_ref is not None
              else _fresh_grad(torch.empty(Cout, dtype=torch.float32, device=dev))) if (has_bias and need[2]) else None
        nbytes = lib.rdst_conv_bwd_workspace(B, H, W, Cin, Cout, k)

        def call(dx_, dw_, db_, wsp_):
            _lib.check(lib.rdst_conv_bwd(x.data_ptr(), ldx, in_act, w.data_ptr(), dy_r.data_ptr(), lddy, _ptr(dx_), Cin,
                                         None, 0, _ptr(dw_), _ptr(db_), wsp_.data_ptr(), nbytes, B, H, W, Cin, Cout, k,
                                         out_scale, r, code, _stream()), "rdst_conv_bwd")

        if TWO_STREAM_BACKWARD and dx is not None and (dw is not None or db is not None):
            cur, side = torch.cuda.current_stream(), _side_stream(dev)
            wsp_w = _workspace(nbytes, dev)
            wsp_d = _workspace(nbytes, dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                call(None, dw, db, wsp_w)
            call(dx, None, None, wsp_d)
            cur.wait_stream(side)
        else:
            call(dx, dw, db, _workspace(nbytes, dev))
        _ReduceBatch.settle(lib)   # (a conv inside an open outer batch: nothing of its own is deferred, its fresh destinations settle)
        dres = dy if (has_res and need[3]) else None
        return dx, dw, db, dres, None, None, None, None


def conv_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], *, in_act: int = ACT_NONE,
              residual: Optional[torch.Tensor] = None, out_scale: float = 1.0, shuffle: int = 1, out_slot=None) -> torch.Tensor:
    """k x k conv (k = 1, 3; zero pad k//2) on token-major x (B,H,W,Cin) with nn.Conv2d weights
    (Cout,Cin,k,k): y = (conv(in_act(x)) + bias) * out_scale + residual, PixelShuffle(shuffle) folded
    into the store -> (B, H*r, W*r, Cout/r^2).  See include/rdst_hip.h, K4/K5/K6."""
    return _ConvRows.apply(x, weight, bias, residual, in_act, out_scale, shuffle, out_slot)


# ------------------------------------------------------------------------------------------------
# NCHW boundary of the module
# ------------------------------------------------------------------------------------------------
class _NchwToRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        _need_gpu(x)
        lib = _lib.load()
        B, C, H, W = x.shape
        xs = x.detach().float().contiguous()
        rows = torch.empty((B, H, W, C), dtype=dtype, device=x.device)
        _lib.check(lib.rdst_nchw_to_rows(xs.data_ptr(), rows.data_ptr(), C, B, C, H, W, _elt_code(rows), _stream()),
                   "rdst_nchw_to_rows")
        return rows

    @staticmethod
    def backward(ctx, drows):
        return _RowsToNchw.apply(drows), None


class _RowsToNchw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows):
        _need_gpu(rows)
        lib = _lib.load()
        B, H, W, C = rows.shape
        r, ld = _rows(rows)
        ctx.dtype = rows.dtype
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=rows.device)
        _lib.check(lib.rdst_rows_to_nchw(r.data_ptr(), ld, out.data_ptr(), B, C, H, W, _elt_code(rows), _stream()),
                   "rdst_rows_to_nchw")
        return out

    @staticmethod
    def backward(ctx, dout):
        return _NchwToRows.apply(dout, ctx.dtype)


class _Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows):
        _need_gpu(rows)
        lib = _lib.load()
        B, H, W, C = rows.shape
        r, ld = _rows(rows)
        out = torch.empty((B, 2 * H, 2 * W, C), dtype=rows.dtype, device=rows.device)
        _lib.check(lib.rdst_upsample2_fwd(r.data_ptr(), ld, out.data_ptr(), C, B, H, W, C, _elt_code(rows), _stream()),
                   "rdst_upsample2_fwd")
        ctx.geom = (B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        B, H, W, C = ctx.geom
        d, ld = _rows(dy)
        dx = torch.empty((B, H, W, C), dtype=dy.dtype, device=dy.device)
        _lib.check(lib.rdst_upsample2_bwd(d.data_ptr(), ld, dx.data_ptr(), C, B, H, W, C, _elt_code(dy), _stream()),
                   "rdst_upsample2_bwd")
        return dx


def upsample_nearest2(rows: torch.Tensor) -> torch.Tensor:
    """token-major rows (B,H,W,C) -> (B,2H,2W,C): F.interpolate(scale_factor=2, mode='nearest') of the reference's
    'nearest+conv' reconstruction (networks/swin_transformer_sr.py:801-802)."""
    return _Upsample2.apply(rows)


def nchw_to_rows(x: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """fp32 NCHW image -> token-major rows (B,H,W,C) of `dtype` (PatchEmbed's flatten+transpose,
    networks/swin_transformer_sr.py:515-516, done once at the module boundary)."""
    return _NchwToRows.apply(x, dtype)


def rows_to_nchw(rows: torch.Tensor) -> torch.Tensor:
    """token-major rows (B,H,W,C) -> fp32 NCHW (PatchUnEmbed, networks/swin_transformer_sr.py:552-555)."""
    return _RowsToNchw.apply(rows)
