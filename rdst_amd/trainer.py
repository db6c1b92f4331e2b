"""Data-parallel training-step shell: the inner loop of the reference trainer
(models/trans_sr_trainer.py:141-173) around the HIP hot path, plus the reference's checkpoint layout.

    forward -> loss -> [loss-threshold guard] -> zero_grad -> backward -> (grad all-reduce) -> Adam -> scheduler

What changes against the reference loop is only what data parallelism and the flat buffers need:
  * gradients live in one ``FlatGradBucket`` (one RCCL all-reduce per step, rdst_amd/dp.py);
  * the optimizer is ``FlatAdam`` (one HIP launch per step, rdst_amd/optim.py) with the reference's
    hyper-parameters (utils/optim.py:30-53) and scheduler (utils/optim.py:56-75);
  * the loss-threshold guard (trans_sr_trainer.py:162, ``loss.item() < loss_threshold``) costs a host
    sync per step; it is evaluated only when the threshold is below ``GUARD_OFF`` (the shipped ini sets
    1e8, i.e. "never skip"), otherwise the step never leaves the device.
``save_checkpoint`` / ``load_checkpoint`` use the key layout of models/basic_trainer.py:164-208
(``model_g``, ``optimizer_g``, ``scheduler_g``, ``loss`` state dicts + the training-state fields), so a
reference ``checkpoint.tar`` resumes here and vice versa (FlatAdam keeps torch.optim.Adam's state layout).
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import dp
from .optim import FlatAdam

GUARD_OFF = 1e8   # config_files/RDST_E1_OASIS_example_SRx4.ini:136


class DPTrainStep:
    """``loss_fn``: a callable ``(pred, target) -> loss`` (default L1) or an ``SRLoss``-shaped object returning
    ``(loss, report)`` (rdst_amd.loss.SRLoss: the weighted 'L1' / 'UNet-F' states of the reference trainer).
    ``graph=True``: after ``graph_warmup`` eager steps on one input shape, forward + loss + backward are captured into ONE
    HIP graph and every later step of that shape replays it (the collective and the optimizer stay outside, so RCCL
    keeps its own streams); other shapes, and steps under an active loss-threshold guard, run eagerly.  This is the
    step ``bench.py`` times."""

    RECORD_FLUSH = 4096   # parked device scalars before they are converted in one batch (bounds the memory they hold)

    def __init__(self, net: torch.nn.Module, lr: float = 1e-4, betas=(0.9, 0.99), eps: float = 1e-8,
                 weight_decay: float = 0.0, milestones: Optional[Sequence[int]] = None, gamma: float = 0.5,
                 loss_threshold: float = GUARD_OFF, loss_fn: Optional[Callable] = None, group=None,
                 graph: bool = False, graph_warmup: int = 2):
        self.net = net
        self.group = group
        dp.broadcast_parameters(net, group=group)
        self.bucket = dp.FlatGradBucket(net.parameters())
        self.optimizer = FlatAdam(self.bucket.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                  bucket=self.bucket)
        self.scheduler = (torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=list(milestones), gamma=gamma)
                          if milestones else None)
        self.loss_fn = loss_fn if loss_fn is not None else F.l1_loss      # SRLoss 'L1' (loss/sr_loss.py)
        # what the reference saves under 'loss' (basic_trainer.py:195): the loss object's own state_dict()
        self.loss_module = loss_fn if hasattr(loss_fn, "state_dict") else torch.nn.Module()
        self.loss_threshold = float(loss_threshold)
        self.last_report = None        # the (lazy) per-component report of the last step, SRLoss-shaped losses only
        # graph capture
        self.use_graph = bool(graph)
        self.graph_warmup = int(graph_warmup)
        self.graph = None
        self._graph_report = None      # the report object whose tensors the captured graph rewrites on every replay
        self._static = None            # (inputs, targets) the graph reads
        self._eager_seen = 0
        self._loss_buf = None
        self.capture_hook = None       # optional context-manager factory wrapped around the capture (bench.py's Recorder)
        # training state carried by the reference's checkpoints
        self.training_loss_names = ["L1"] if loss_fn is None else list(getattr(loss_fn, "loss_components", ["loss"]))
        self.training_loss_records: Dict[str, list] = {n: [] for n in self.training_loss_names}
        self._pending: Dict[str, list] = {}      # device scalars of the steps since the last flush (see _record)
        self.quick_validation_reports: list = []
        self.current_training_state_id = 0
        self.current_epoch = 0
        self.training_epoch_costs: list = []

    def _keep_step(self, loss: torch.Tensor) -> bool:
        """The loss-threshold decision of trans_sr_trainer.py:162, made COLLECTIVELY: every rank must take the same
        branch, or the ranks that skip would leave the others waiting in the gradient all-reduce.  The group skips
        when ANY rank's local loss is not below the threshold (MAX over ranks of the "skip" flag; a NaN loss skips)."""
        skip = torch.logical_not(loss.detach() < self.loss_threshold).to(torch.float32).reshape(1)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(skip, op=dist.ReduceOp.MAX, group=self.group)
        return float(skip.item()) == 0.0

    def _loss(self, out, targets):
        r = self.loss_fn(out, targets)
        if isinstance(r, tuple):
            self.last_report = r[1]
            return r[0]
        return r

    def _record(self, loss: torch.Tensor) -> None:
        """trans_sr_trainer.py:165-167 appends ``report[name]`` (a Python float: one host sync per component and step).
        Here the DEVICE scalars of the step are parked (a clone each: the graph step overwrites its buffers) and become
        floats only when a checkpoint is written (`_flush_records`), so a step never leaves the device."""
        rep = self.last_report
        if isinstance(rep, dict) and rep:
            for n in self.training_loss_names:
                if n in rep:
                    v = rep.raw(n) if hasattr(rep, "raw") else dict.__getitem__(rep, n)
                    self._pending.setdefault(n, []).append(v.detach().clone() if torch.is_tensor(v) else float(v))
        elif len(self.training_loss_names) == 1:
            self._pending.setdefault(self.training_loss_names[0], []).append(loss.detach().clone())
        if sum(len(v) for v in self._pending.values()) >= self.RECORD_FLUSH:
            self._flush_records()

    def _flush_records(self) -> None:
        for n, vals in self._pending.items():
            if not vals:
                continue
            ts = [v for v in vals if torch.is_tensor(v)]
            host = torch.stack([t.reshape(()).float() for t in ts]).cpu().tolist() if ts else []
            it = iter(host)
            self.training_loss_records.setdefault(n, []).extend(next(it) if torch.is_tensor(v) else v for v in vals)
        self._pending = {}

    def loss_records(self) -> Dict[str, list]:
        """``training_loss_records`` with every parked step converted (the reference reads the attribute directly,
        basic_trainer.py:440; here the attribute lags by up to RECORD_FLUSH steps until this or checkpoint() runs)."""
        self._flush_records()
        return self.training_loss_records

    def fwd_bwd(self, inputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """forward + loss + backward with the gradients written straight into the flat bucket (no host sync inside:
        capturable).  Returns the loss as a device scalar that is overwritten by the next call."""
        self.bucket.detach_grads()       # autograd assigns fresh gradients; the HIP ops take the bucket views as destinations
        out = self.net(inputs)
        loss = self._loss(out, targets)
        if self._loss_buf is None:
            self._loss_buf = torch.zeros((), dtype=torch.float32, device=inputs.device)
        self._loss_buf.copy_(loss.detach())
        # (One reduction batch around the WHOLE backward — 6 + 3 slab-sum launches instead of 24 + 24 — was measured in round 4:
        # 17.6 -> 20.0 ms/step.  The ~1.7 GB of slabs then stay allocated until the end of the pass; per DenseSTLayer they are
        # 22-30 MB buffers that the caching allocator hands out again and again, written and summed while still in the 256 MiB
        # Infinity Cache.)
        loss.backward()
        self.bucket.gather()             # whatever was not written in place is flattened into the bucket
        return self._loss_buf

    def capture(self, inputs: torch.Tensor, targets: torch.Tensor) -> bool:
        """Capture fwd_bwd on (copies of) these tensors into a HIP graph.  False (and eager from then on) if the capture
        fails; the parameters and BatchNorm statistics are not touched by a failed capture."""
        import contextlib
        self._static = (inputs.detach().clone(), targets.detach().clone())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with (self.capture_hook() if self.capture_hook is not None else contextlib.nullcontext()):
                with torch.cuda.graph(g):
                    self.fwd_bwd(*self._static)
        except Exception as e:  # noqa: BLE001 - fall back to eager, loudly
            import warnings
            warnings.warn(f"rdst_amd.trainer: HIP-graph capture failed ({type(e).__name__}: {e}); running eagerly")
            torch.cuda.synchronize()
            # a backward that died half way leaves a reduction batch open and bucket views on offer: drop both (the
            # queued reductions name workspaces of the failed capture; running them later would write through freed memory)
            # (the batch lives on autograd's device thread: reset_backward_state() reaches it through an epoch counter)
            from . import ops
            ops.reset_backward_state()
            dp._OFFERED = {}
            self.bucket.gather()
            self.use_graph, self.graph, self._static, self._graph_report = False, None, None, None
            return False
        if not self.bucket.check_views():
            self.use_graph, self.graph, self._static, self._graph_report = False, None, None, None
            return False
        self.graph = g
        # the report the capture produced: its tensors are the graph's own buffers, refreshed by every replay.  An eager
        # step in between (another shape) rebinds last_report to ITS tensors; step() re-points it before recording a replay.
        self._graph_report = self.last_report
        return True

    def _graph_fits(self, inputs, targets) -> bool:
        return (self._static is not None and inputs.shape == self._static[0].shape and targets.shape == self._static[1].shape
                and inputs.dtype == self._static[0].dtype and targets.dtype == self._static[1].dtype)

    def step(self, inputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """One iteration of trans_sr_trainer.py:131-178; returns the (device) loss."""
        t0 = time.time()                               # :132
        self.current_epoch += 1                        # :134 — advanced whether or not the update is skipped
        self.net.train()
        guarded = self.loss_threshold < GUARD_OFF
        if guarded:
            # the reference's order: forward, loss, decide, then backward (trans_sr_trainer.py:149-174); one host sync
            self.bucket.detach_grads()
            out = self.net(inputs)
            loss = self._loss(out, targets)
            if self._keep_step(loss):                  # :162
                self._record(loss)                     # :165-167
                loss.backward()
                self.bucket.gather()
                self._finish_step()
            else:
                self.bucket.gather()                   # restore p.grad = bucket views (zeros for this step)
            self.training_epoch_costs.append(time.time() - t0)
            return loss.detach()
        if self.use_graph and self.graph is None and self._eager_seen >= self.graph_warmup:
            self.capture(inputs, targets)
        if self.graph is not None and self._graph_fits(inputs, targets):
            if inputs.data_ptr() != self._static[0].data_ptr():
                self._static[0].copy_(inputs)
            if targets.data_ptr() != self._static[1].data_ptr():
                self._static[1].copy_(targets)
            self.graph.replay()
            self.last_report = self._graph_report
            loss = self._loss_buf
        else:
            loss = self.fwd_bwd(inputs, targets)
            self._eager_seen += 1
        self._record(loss)                             # :165-167 (lazy: no host sync)
        self._finish_step()
        self.training_epoch_costs.append(time.time() - t0)   # :176-178 (host-side enqueue time: nothing synced)
        return loss

    def _finish_step(self) -> None:
        self.bucket.all_reduce_mean(self.group)        # no-op on one rank
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()

    # ---- models/basic_trainer.py:164-208 -----------------------------------------------------------
    def checkpoint(self) -> dict:
        self._flush_records()
        ck = {"Time": time.strftime("%Y-%m-%d %H:%M:%S"),
              "model_g": self.net.state_dict(),
              "optimizer_g": self.optimizer.state_dict(),
              "loss": self.loss_module.state_dict(),
              "training_loss_names": self.training_loss_names,
              "training_loss_records": self.training_loss_records,
              "quick_validation_reports": self.quick_validation_reports,
              "current_training_state_id": self.current_training_state_id,
              "current_epoch": self.current_epoch,
              "training_epoch_costs": self.training_epoch_costs}
        if self.scheduler is not None:
            ck["scheduler_g"] = self.scheduler.state_dict()
        return ck

    def save_checkpoint(self, path: str) -> None:
        torch.save(self.checkpoint(), path)

    def load_checkpoint(self, path_or_dict, map_location=None) -> None:
        ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location=map_location,
                                                                            weights_only=False)
        # parameters are views of the optimizer's flat buffer: load_state_dict copies in place and keeps them
        self.net.load_state_dict(ck["model_g"])
        self.optimizer.load_state_dict(ck["optimizer_g"])
        if self.scheduler is not None and "scheduler_g" in ck:
            self.scheduler.load_state_dict(ck["scheduler_g"])
        if "loss" in ck and len(ck["loss"]) and hasattr(self.loss_module, "load_state_dict"):
            self.loss_module.load_state_dict(ck["loss"])
        self.training_loss_names = ck.get("training_loss_names", self.training_loss_names)
        self.training_loss_records = ck.get("training_loss_records", self.training_loss_records)
        self._pending = {}             # steps taken before the load belong to the history that was just replaced
        self.quick_validation_reports = ck.get("quick_validation_reports", [])
        self.current_training_state_id = ck.get("current_training_state_id", 0)
        self.current_epoch = ck.get("current_epoch", 0)
        self.training_epoch_costs = ck.get("training_epoch_costs", [])
