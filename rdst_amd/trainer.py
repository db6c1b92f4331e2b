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
    def __init__(self, net: torch.nn.Module, lr: float = 1e-4, betas=(0.9, 0.99), eps: float = 1e-8,
                 weight_decay: float = 0.0, milestones: Optional[Sequence[int]] = None, gamma: float = 0.5,
                 loss_threshold: float = GUARD_OFF, loss_fn: Optional[Callable] = None, group=None):
        self.net = net
        self.group = group
        dp.broadcast_parameters(net, group=group)
        self.bucket = dp.FlatGradBucket(net.parameters())
        self.optimizer = FlatAdam(self.bucket.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                  bucket=self.bucket)
        self.scheduler = (torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=list(milestones), gamma=gamma)
                          if milestones else None)
        self.loss_fn = loss_fn if loss_fn is not None else F.l1_loss      # SRLoss 'L1' (loss/sr_loss.py)
        self.loss_module = loss_fn if isinstance(loss_fn, torch.nn.Module) else torch.nn.Module()
        self.loss_threshold = float(loss_threshold)
        # training state carried by the reference's checkpoints
        self.training_loss_names = ["L1"] if loss_fn is None else ["loss"]
        self.training_loss_records: Dict[str, list] = {n: [] for n in self.training_loss_names}
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

    def step(self, inputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """One iteration of trans_sr_trainer.py:131-178; returns the (device) loss."""
        t0 = time.time()                               # :132
        self.current_epoch += 1                        # :134 — advanced whether or not the update is skipped
        self.net.train()
        out = self.net(inputs)
        loss = self.loss_fn(out, targets)
        if self.loss_threshold >= GUARD_OFF or self._keep_step(loss):   # :162
            self.optimizer.zero_grad()                 # one memset of the flat bucket
            loss.backward()
            self.bucket.all_reduce_mean(self.group)    # no-op on one rank
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
        self.training_epoch_costs.append(time.time() - t0)   # :176-178 (host-side enqueue time unless the guard synced)
        return loss.detach()

    # ---- models/basic_trainer.py:164-208 -----------------------------------------------------------
    def checkpoint(self) -> dict:
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
        if "loss" in ck and len(ck["loss"]) and isinstance(self.loss_module, torch.nn.Module):
            self.loss_module.load_state_dict(ck["loss"])
        self.training_loss_names = ck.get("training_loss_names", self.training_loss_names)
        self.training_loss_records = ck.get("training_loss_records", self.training_loss_records)
        self.quick_validation_reports = ck.get("quick_validation_reports", [])
        self.current_training_state_id = ck.get("current_training_state_id", 0)
        self.current_epoch = ck.get("current_epoch", 0)
        self.training_epoch_costs = ck.get("training_epoch_costs", [])
