"""Seg-UNet perceptual loss in the backward path (BASELINE.json configs[4]):
the reference's ``SegUNet_F`` (loss/seg_unet.py) for its 'encoder' modes on loss layers 0 and 1.

What the reference does (loss/seg_unet.py:46-107): a ``segmentation_models_pytorch.Unet(in_channels, classes=4)`` (resnet34
encoder) with weights from ``loss/unet_*.pt`` computes features of the SR output (with grad) and of the HR image (no_grad);
``encoder`` modes compare ``features[l]`` for l in ``loss_layers``; features[0] is the image itself, features[1] the stem
``relu(bn1(conv1(x)))`` at half resolution.  Quirks kept: a mode name containing 'L1' selects **MSELoss** and 'L2' selects
L1Loss (:73-76); ``loss /= len(layers)`` sits INSIDE the loop (:105-107); the UNet is never put in ``eval()``, so BatchNorm
uses batch statistics and updates its running statistics on the SR batch and on the HR batch of every call.

**Parity unpinned**: neither segmentation_models_pytorch nor the weight files exist in the build image,
so there is nothing to generate golden vectors from.  The stem here has the state-dict keys of smp's resnet34 encoder
(``encoder.conv1.weight``, ``encoder.bn1.*``): a reference ``unet_oasis.pt`` loads with ``strict=False`` and its stem
is used; without a file the stem keeps torchvision's ResNet initialisation.  Deeper layers, the decoder and the 'label'
(Dice, = RDST-HRL) modes need the whole smp UNet and are not built: NotImplementedError.

The stem itself (conv 7x7 / 2 -> BatchNorm(train) -> ReLU -> MSE / L1, forward and backward to the SR image) is one pair
of HIP entry points (rdst_stem_loss_fwd / _bwd, csrc/stem_loss.hip); there is no CPU fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib


class _StemLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sr, hr, conv_w, bn_w, bn_b, rmean, rvar, momentum, eps, use_mse):
        if not sr.is_cuda:
            raise RuntimeError("rdst_amd.loss.SegUNet_F: the stem loss is a HIP kernel; there is no CPU fallback")
        lib = _lib.load()
        B, Cin, H, W = sr.shape
        srs, hrs = sr.detach().float().contiguous(), hr.detach().float().contiguous()
        nb = lib.rdst_stem_loss_workspace(B, H, W)
        wsp = torch.empty(nb, dtype=torch.uint8, device=sr.device)
        loss = torch.empty((), dtype=torch.float32, device=sr.device)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.rdst_stem_loss_fwd(srs.data_ptr(), hrs.data_ptr(), conv_w.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                          rmean.data_ptr() if rmean is not None else None,
                                          rvar.data_ptr() if rvar is not None else None, float(momentum), float(eps),
                                          int(use_mse), loss.data_ptr(), wsp.data_ptr(), nb, B, Cin, H, W, st), "rdst_stem_loss_fwd")
        ctx.save_for_backward(conv_w, bn_w, wsp)
        ctx.geom = (B, Cin, H, W, nb)
        return loss

    @staticmethod
    def backward(ctx, gout):
        conv_w, bn_w, wsp = ctx.saved_tensors
        B, Cin, H, W, nb = ctx.geom
        lib = _lib.load()
        up = gout.detach().float().contiguous()
        dsr = torch.empty((B, Cin, H, W), dtype=torch.float32, device=wsp.device)
        _lib.check(lib.rdst_stem_loss_bwd(conv_w.data_ptr(), bn_w.data_ptr(), up.data_ptr(), dsr.data_ptr(), wsp.data_ptr(), nb,
                                          B, Cin, H, W, torch.cuda.current_stream().cuda_stream), "rdst_stem_loss_bwd")
        return dsr, None, None, None, None, None, None, None, None, None


class _Stem(nn.Module):
    """conv1 + bn1 of smp's resnet34 encoder, under its state-dict names."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        nn.init.kaiming_normal_(self.conv1.weight, mode="fan_out", nonlinearity="relu")   # torchvision ResNet.__init__


class SegUNet_F(nn.Module):
    _MODES = {"OASIS": (1, "loss/unet_oasis.pt"), "BraTS": (4, "loss/unet_brats.pt"), "ACDC": (1, "loss/unet_acdc.pt"),
              "COVID": (1, "loss/unet_covid.pt")}

    def __init__(self, loss_layers: Dict[str, list], mode: str = "OASIS", unet_path: Optional[str] = None):
        super().__init__()
        in_channels = None
        for k, (c, path) in self._MODES.items():
            if k in mode:
                in_channels, default_path = c, path
        if in_channels is None:
            raise ValueError("Invalid UNet Seg Loss data mode: {}".format(mode))
        for k in loss_layers:                         # loss/seg_unet.py:51-53: the last key wins
            self.loss_mode = k
        self.loss_layers = list(loss_layers[self.loss_mode])
        if "encoder" not in self.loss_mode:
            raise NotImplementedError(
                "rdst_amd.loss.SegUNet_F: only the 'encoder' modes are built (the decoder / label modes need the whole "
                "segmentation_models_pytorch UNet, which is outside this repository's scope: SURVEY.md section 8f N2)")
        if any(l not in (0, 1) for l in self.loss_layers):
            raise NotImplementedError("rdst_amd.loss.SegUNet_F: encoder loss layers 0 (image) and 1 (stem) are built; "
                                      "deeper layers need the resnet34 body")
        self.encoder = _Stem(in_channels)
        path = unet_path if unet_path is not None else default_path
        import os
        if os.path.exists(path):                      # a reference UNet checkpoint: take its stem
            sd = torch.load(path, map_location="cpu")
            self.load_state_dict({k: v for k, v in sd.items() if k.startswith(("encoder.conv1.", "encoder.bn1."))}, strict=False)
        self.loss_names = ["SegUNet({})".format(self.loss_mode)]
        self.use_mse = "L1" in self.loss_mode         # loss/seg_unet.py:73-78: 'L1' -> MSELoss, everything else L1Loss

    def _pixel_loss(self, a, b):
        return F.mse_loss(a, b) if self.use_mse else F.l1_loss(a, b)

    def forward(self, sr, hr, gt_label=None):
        assert sr.shape == hr.shape, "Seg UNet Loss invalid SR({}) and HR({}) shape!".format(sr.shape, hr.shape)
        bn = self.encoder.bn1
        loss = 0
        for l in self.loss_layers:
            if l == 0:
                term = self._pixel_loss(sr, hr.detach())
            else:
                term = _StemLoss.apply(sr, hr, self.encoder.conv1.weight.detach(), bn.weight.detach(), bn.bias.detach(),
                                       bn.running_mean, bn.running_var, bn.momentum, bn.eps, self.use_mse)
                bn.num_batches_tracked += 2           # one BatchNorm forward on SR, one on HR
            loss = loss + term
            loss = loss / len(self.loss_layers)       # :105-107: inside the loop, as the reference has it
        return loss, {self.loss_names[0]: loss.item()}
