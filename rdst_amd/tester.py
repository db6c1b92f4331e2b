"""Whole-slice inference shell: the inner loop of the reference tester
(models/trans_sr_tester.py:124-166) around the HIP network, plus the scoring of metrics/sr_evaluation.py:143-156.

    net.eval(); with torch.no_grad(): for p in lr_img.split(batch_size * 4): rec.append(net(p)); rec = torch.cat(rec)

The network runs on any H x W that is a multiple of the window size (non-square, other than the constructor's img_size)
and any batch size; the shifted-window mask is analytic, so nothing is rebuilt per shape."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

from .metrics import SRMetrics


class SRTester:
    def __init__(self, net: torch.nn.Module, batch_size: int = 16, sr_scale: Optional[float] = None,
                 metrics: str = "psnr ssim", compute_dtype: Optional[torch.dtype] = None):
        self.net = net
        self.batch_size = int(batch_size)
        self.sr_scale = float(sr_scale if sr_scale is not None else getattr(net, "sr_scale", getattr(net, "upscale", 1)))
        self.metrics = SRMetrics(metrics, "full")
        if compute_dtype is not None and hasattr(net, "set_compute_dtype"):
            net.set_compute_dtype(compute_dtype)

    @torch.no_grad()
    def inference(self, lr_img: torch.Tensor) -> torch.Tensor:
        """trans_sr_tester.py:141-160: (N, C, H, W) low-resolution slices -> (N, C, H*s, W*s)."""
        self.net.eval()
        dev = next(self.net.parameters()).device
        rec = [self.net(p.to(dev)) for p in lr_img.split(self.batch_size * 4)]
        return torch.cat(rec, dim=0)

    def evaluate(self, lr_img: torch.Tensor, gt_img: torch.Tensor) -> Dict[str, List[float]]:
        """Scores of metrics/sr_evaluation.py:152: per-image metrics after cropping ceil(sr_scale) border pixels."""
        rec = self.inference(lr_img)
        return self.metrics(gt_img, rec, int(math.ceil(self.sr_scale)))
