"""Pure data parallelism for the RDST hot path: one process per GPU, one flat gradient bucket, one
RCCL all-reduce per step.

The reference has no distributed code at all (``# do distributed training here`` is its only trace,
train.py:29); patches are independent in forward and backward, so the only exchange is the
parameter gradient: 4,464,961 fp32 values = 17.9 MB for RDST-E1.  xGMI is
point-to-point and this payload is latency-bound, so it goes out as ONE collective over ONE
contiguous buffer instead of 750 small ones: every ``param.grad`` is a view into the bucket, so
autograd accumulates straight into it and no gather/scatter copy exists.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


# A one-rank process group normally skips its collectives (nothing to exchange).  bench.py --force-pg sets this so that the
# SAME calls the 8-GPU run makes (parameter broadcast, AVG all-reduce of the flat bucket) execute on RCCL with one rank: the
# transport of row (e) can be exercised on a one-GPU box.
FORCE_COLLECTIVES = False

# data_ptr of a parameter -> its bucket view, while a detach_grads()/gather() bracket is open (see FlatGradBucket)
_OFFERED: dict = {}


def take_grad_view(param: torch.Tensor):
    """The bucket view that should receive this parameter's gradient, or None.  Each view is handed out once per
    backward pass (a parameter used twice gets a fresh tensor the second time and autograd accumulates)."""
    return _OFFERED.pop(param.data_ptr(), None) if _OFFERED else None


class FlatGradBucket:
    """Owns one contiguous fp32 buffer; ``p.grad`` of every trainable parameter is a view of it.

    Two ways to fill it:
      * ``zero()`` then ``backward()``: autograd ACCUMULATES into the views (one small add kernel per
        parameter, 750 for RDST-E1);
      * ``detach_grads()`` then ``backward()`` then ``gather()``: autograd ASSIGNS fresh gradient tensors
        (no adds, no memset) and ``gather()`` flattens them into the bucket with a handful of batched copy
        kernels and re-points ``p.grad`` at the views.  This is what bench.py captures into its HIP graph."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket: no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatGradBucket: parameters must be fp32 on one device")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def zero(self) -> None:
        """One memset instead of 750 (replaces optimizer.zero_grad(), trans_sr_trainer.py:170)."""
        self.flat.zero_()

    def detach_grads(self) -> None:
        """Forget the views: the next backward() assigns new gradient tensors instead of accumulating.
        The bucket views are offered to the HIP ops as the destination of those gradients (``take_grad_view``):
        a weight-gradient kernel then writes straight into the bucket and ``gather()`` has nothing to copy."""
        global _OFFERED
        offered = {}
        off = 0
        for p in self.params:
            p.grad = None
            offered[p.data_ptr()] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        _OFFERED = offered

    def gather(self) -> None:
        """Flatten the freshly assigned gradients into the bucket (parameters the loss does not reach get
        zeros) and make ``p.grad`` the bucket views again."""
        global _OFFERED
        _OFFERED = {}
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            g = p.grad
            if g is None:
                view.zero_()
            elif g.data_ptr() != base + 4 * off:     # not written in place by the op: copy it in
                view.copy_(g)
            p.grad = view
            off += p.numel()

    def check_views(self) -> bool:
        """True while every p.grad still aliases the bucket (zero_grad(set_to_none=True) breaks it)."""
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return False
            off += p.numel()
        return True

    def all_reduce_mean(self, group=None) -> None:
        """Average the bucket over the ranks: one collective (RCCL over xGMI on the 'nccl' backend)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1 and not FORCE_COLLECTIVES:
            return
        if dist.get_backend(group) == "nccl":      # RCCL averages inside the collective: one kernel fewer
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
        else:                                      # gloo has no AVG
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.mul_(1.0 / world)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers (one flat broadcast each dtype)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES):
        return
    tensors = [p.data for p in module.parameters()] + [b for b in module.buffers() if b is not None]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
