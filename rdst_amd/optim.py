"""Fused Adam for the data-parallel training step.

``FlatAdam`` is a ``torch.optim.Optimizer`` with torch.optim.Adam's update rule — the optimizer the
reference builds in utils/optim.py:30-53 (Adam, betas (0.9, 0.99), eps 1e-8, lr 1e-4, weight_decay 0;
config_files/RDST_E1_OASIS_example_SRx4.ini:128-135) and steps in models/trans_sr_trainer.py:170-173 —
but parameters, gradients and both moments each live in ONE contiguous fp32 buffer and a step is ONE
HIP launch (``rdst_adam_step``) instead of a multi-tensor sweep per moment plus per-parameter
bookkeeping kernels.  Being an ``Optimizer`` it works with ``torch.optim.lr_scheduler`` (the
reference's MultiStepLR, utils/optim.py:56-75) and its ``state_dict()`` has torch.optim.Adam's layout,
so the optimizer part of a reference ``checkpoint.tar`` (models/basic_trainer.py:187-208) loads.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import _lib
from .dp import FlatGradBucket


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas=(0.9, 0.99), eps: float = 1e-8,
                 weight_decay: float = 0.0, bucket: Optional[FlatGradBucket] = None):
        params = [p for p in params if p.requires_grad]
        if bucket is not None and [id(p) for p in bucket.params] != [id(p) for p in params]:
            raise ValueError("FlatAdam: the gradient bucket must hold exactly these parameters, in this order")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam: one parameter group only")
        self.bucket = bucket if bucket is not None else FlatGradBucket(params)
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._steps = 0
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)  # the module now computes from the flat buffer
                off += k
        self._point_state()

    def _point_state(self) -> None:
        # ONE host-side step counter shared by all 750 per-parameter states (torch.optim.Adam's layout wants a
        # "step" entry per parameter; they are always equal here): a step() updates it once, not 750 times
        self._step_t = torch.tensor(float(self._steps))
        off = 0
        for p in self.param_groups[0]["params"]:
            k = p.numel()
            self.state[p] = {"step": self._step_t,
                             "exp_avg": self.exp_avg[off:off + k].view_as(p),
                             "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p)}
            off += k

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.flat_param.is_cuda:
            raise RuntimeError("rdst_amd.optim.FlatAdam.step: the fused update is a HIP kernel; there is no CPU fallback")
        if not self.bucket.check_views():
            raise RuntimeError("FlatAdam.step: p.grad no longer aliases the gradient bucket "
                               "(use bucket.zero() / bucket.detach_grads()+gather(), not zero_grad(set_to_none=True))")
        g = self.param_groups[0]
        self._steps += 1
        lib = _lib.load()
        _lib.check(lib.rdst_adam_step(self.flat_param.data_ptr(), self.bucket.flat.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self.flat_param.numel(), float(g["lr"]),
                                      float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                      float(g["weight_decay"]), self._steps,
                                      torch.cuda.current_stream().cuda_stream), "rdst_adam_step")
        self._step_t.fill_(float(self._steps))
        return loss

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: D401 - keeps p.grad aliased to the bucket
        self.bucket.zero()

    def load_state_dict(self, state_dict) -> None:
        """Accepts a torch.optim.Adam state dict (same layout as ours) and re-flattens it."""
        super().load_state_dict(state_dict)
        off = 0
        steps = 0
        with torch.no_grad():
            for p in self.param_groups[0]["params"]:
                k = p.numel()
                st = self.state.get(p, {})
                if "exp_avg" in st:
                    self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    steps = max(steps, int(float(st["step"])))
                else:
                    self.exp_avg[off:off + k].zero_()
                    self.exp_avg_sq[off:off + k].zero_()
                off += k
        self._steps = steps
        self._point_state()
