"""The weighted training loss of the reference trainer (loss/sr_loss.py:11-51, loss/basic_loss.py:14-95) for the loss
components that sit on this repository's path: 'L1' / 'L2' / 'MSE' (``RecLoss``, sr_loss.py:60-72) and 'UNet-F'
(``SegUNet_F``, loss/seg_unet.py).  VGG / GAN components are outside SURVEY.md section 8 and raise.

    loss, repo = SRLoss(paras)(pred, gt)       # sum_n scalars[state][n] * component_n(pred, gt)

What differs from the reference is only what a graph-captured, sync-free training step needs:
  * the per-component report holds DEVICE scalars and converts them on access (``LazyScalars``): the reference calls
    ``loss.item()`` inside every component (sr_loss.py:72, seg_unet.py:127), one host sync each — a synced step cannot be
    captured into a HIP graph.  Reading ``repo[name]`` still yields a Python float, as the reference's does;
  * ``paras.unet_path`` (optional) overrides the UNet checkpoint path, ``paras.unet_allow_random`` (tests / benchmarks
    on synthetic data) lets the UNet keep its seeded random initialisation when no checkpoint exists.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .seg_unet import SegUNet_F


class LazyScalars(dict):
    """{name: 0-dim tensor}; item access converts to float (the only place a host sync can happen)."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return float(v) if isinstance(v, torch.Tensor) else v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def raw(self, k):
        return dict.__getitem__(self, k)


class RecLoss(object):
    """loss/sr_loss.py:60-72."""

    def __init__(self, type="L1"):
        if type == "L1":
            self.loss_names = ["Rec_L1"]
            self.function = F.l1_loss
        elif type in ["L2", "MSE"]:
            self.loss_names = ["Rec_MSE"]
            self.function = F.mse_loss
        else:
            raise ValueError("Invalid reconstruction loss: {}".format(type))

    def __call__(self, rec, gt):
        loss = self.function(rec, gt)
        return loss, LazyScalars({self.loss_names[0]: loss.detach()})


class SRLoss(object):
    """Reads from ``paras``: training_losses, loss_scalars, training_states (+ unet_loss_layers, unet_loss_mode when
    'UNet-F' is among the losses; gpu_id as the reference's BasicLoss does, -1 = keep modules where they are built)."""

    def __init__(self, paras):
        self.paras = paras
        gpu_id = getattr(paras, "gpu_id", 0)
        self.device = torch.device("cpu") if gpu_id == -1 else torch.device("cuda:{}".format(gpu_id))
        self.precision = getattr(paras, "precision", False)
        self.training_loss_names = list(paras.training_losses)
        self.training_loss_scalars = paras.loss_scalars
        self.current_training_state = paras.training_states[0]
        self.loss_components = []
        self.loss_functions: Dict[str, object] = {}
        self.use_seg_loss_flag = False
        for l in self.training_loss_names:
            if l in ["L1", "L2", "MSE"]:
                f = RecLoss(l)
            elif l in ["UNet-F"]:
                self.use_seg_loss_flag = True
                f = SegUNet_F(paras.unet_loss_layers, paras.unet_loss_mode, unet_path=getattr(paras, "unet_path", None),
                              allow_random_init=bool(getattr(paras, "unet_allow_random", False))).to(self.device)
            else:
                raise NotImplementedError(
                    "rdst_amd.loss.SRLoss: loss component {!r} is outside this repository's scope (SURVEY.md section 8: "
                    "L1 / L2 / MSE and UNet-F are on the path; VGG and GAN losses are not)".format(l))
            self.loss_components += f.loss_names
            self.loss_functions[l] = f

    def __call__(self, pred, gt, sr_scales=None, gt_label=None):
        repo = LazyScalars()
        scalars = self.training_loss_scalars[self.current_training_state]
        loss = 0.
        for n in scalars:
            s = scalars[n]
            f = self.loss_functions[n]
            if "UNet" in n:
                l, r = f(pred, gt, gt_label)
            else:
                l, r = f(pred, gt)
            for k in r:
                dict.__setitem__(repo, k, r.raw(k) if isinstance(r, LazyScalars) else r[k])
            loss = loss + l * s
        return loss, repo

    def set_training_state(self, ts):
        self.current_training_state = ts

    # loss/basic_loss.py:62-88: only nn.Module components carry state
    def state_dict(self):
        return {n: f.state_dict() for n, f in self.loss_functions.items() if isinstance(f, torch.nn.Module)}

    def load_state_dict(self, checkpoint):
        for n, f in self.loss_functions.items():
            if isinstance(f, torch.nn.Module):
                f.load_state_dict(checkpoint[n])

    def apply(self, fn):
        pass    # the reference applies `fn` to GAN components only (sr_loss.py:53-57); none exist here
