"""PSNR / SSIM of super-resolved images the way the reference's tester scores them.

The reference computes both on the HOST with scikit-image (metrics/sr_metrics.py:8-14: ``peak_signal_noise_ratio(GT, P,
data_range=1)`` and ``structural_similarity(GT, P, data_range=1, multichannel=True)``) on (H, W, C) images whose borders
were cropped by ``margin = ceil(sr_scale)`` pixels (metrics/sr_metrics.py:108-115, metrics/sr_evaluation.py:152).  They
stay host-side numpy here too: they are the caller's bookkeeping around the hot path, not part of it.

scikit-image is not installed in the build image, so this is a restatement of the published algorithms:
  * PSNR  = 10 log10(data_range^2 / mean((GT - P)^2)) in float64 — identical to oracle.psnr, which the golden fixtures pin;
  * SSIM  = scikit-image's default (Wang et al. 2004): 7x7 uniform window (scipy.ndimage.uniform_filter, reflect
    borders), K1 = 0.01, K2 = 0.03, sample covariance (N / (N - 1)), mean over the map cropped by (win - 1) / 2 pixels,
    averaged over channels.  **Parity unpinned** (no scikit-image here to generate vectors from); the tests check the
    definition on closed-form cases.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence, Union

import numpy as np

try:  # scipy is available in the image; the pure-numpy path below is the same filter for when it is not
    from scipy.ndimage import uniform_filter as _uniform_filter
except Exception:  # noqa: BLE001
    _uniform_filter = None


def psnr(gt: np.ndarray, pred: np.ndarray, data_range: float = 1.0) -> float:
    """skimage.metrics.peak_signal_noise_ratio (metrics/sr_metrics.py:8-9)."""
    gt = np.asarray(gt, dtype=np.float64)
    pred = np.asarray(pred, dtype=np.float64)
    err = np.mean((gt - pred) ** 2)
    if err == 0:
        return float("inf")
    return float(10.0 * math.log10((data_range ** 2) / err))


def _box(a: np.ndarray, win: int) -> np.ndarray:
    if _uniform_filter is not None:
        return _uniform_filter(a, size=win, mode="reflect")
    pad = win // 2
    ap = np.pad(a, pad, mode="symmetric")          # scipy's 'reflect' = numpy's 'symmetric'
    c = np.cumsum(np.cumsum(np.pad(ap, ((1, 0), (1, 0))), 0), 1)
    H, W = a.shape
    return (c[win:win + H, win:win + W] - c[:H, win:win + W] - c[win:win + H, :W] + c[:H, :W]) / (win * win)


def ssim(gt: np.ndarray, pred: np.ndarray, data_range: float = 1.0, win_size: int = 7) -> float:
    """skimage.metrics.structural_similarity(GT, P, data_range=1, multichannel=True) on (H, W, C) images
    (metrics/sr_metrics.py:12-13)."""
    gt = np.asarray(gt, dtype=np.float64)
    pred = np.asarray(pred, dtype=np.float64)
    if gt.ndim == 2:
        gt, pred = gt[..., None], pred[..., None]
    if gt.shape != pred.shape or min(gt.shape[:2]) < win_size:
        raise ValueError("ssim: images must have the same shape and be at least win_size in both dimensions")
    NP = win_size * win_size
    cov_norm = NP / (NP - 1.0)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    pad = (win_size - 1) // 2
    vals = []
    for c in range(gt.shape[-1]):
        X, Y = gt[..., c], pred[..., c]
        ux, uy = _box(X, win_size), _box(Y, win_size)
        uxx, uyy, uxy = _box(X * X, win_size), _box(Y * Y, win_size), _box(X * Y, win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
        vals.append(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean())
    return float(np.mean(vals))


_FUNCS = {"psnr": psnr, "ssim": ssim}


def _to_hwc_list(imgs, margin: int) -> List[np.ndarray]:
    """SRMetrics.prepare_data (metrics/sr_metrics.py:96-118): tensors (N, C, H, W) / (C, H, W) or arrays (N, H, W, C) /
    (H, W, C), lists of either; borders cropped by ``margin``."""
    import torch
    if isinstance(imgs, (list, tuple)):
        imgs = torch.stack(list(imgs)) if isinstance(imgs[0], torch.Tensor) else np.stack(list(imgs))
    if isinstance(imgs, torch.Tensor):
        a = imgs.detach().float().cpu().numpy()
        a = a.transpose(1, 2, 0) if a.ndim == 3 else a.transpose(0, 2, 3, 1)
    else:
        a = np.asarray(imgs)
    if a.ndim not in (3, 4):
        raise AssertionError("images should have 3 or 4 dimensions")
    H, W = a.shape[-3:-1]
    a = a[..., margin:H - margin, margin:W - margin, :]
    return [a] if a.ndim == 3 else list(a)


class SRMetrics:
    """The pixel metrics of the reference's ``SRMetrics`` that the shipped configs ask for ('psnr ssim',
    config_files/RDST_E1_OASIS_example_SRx4.ini): ``SRMetrics('psnr ssim')(gts, preds, margin)`` -> {'psnr': [...], ...}.
    The other metrics of the reference (sewar's, FID) are outside this repository's scope and raise."""

    def __init__(self, metrics: str = "psnr ssim", return_mode: str = "full"):
        self.metrics = metrics.split()
        for m in self.metrics:
            if m not in _FUNCS:
                raise ValueError("Do not support this metric: {}".format(m))
        if return_mode not in ("full", "mean"):
            raise ValueError("return mode must be one of [mean, full]")
        self.return_mode = return_mode

    def __call__(self, gts, preds, margin: int = 0) -> Dict[str, Union[List[float], float]]:
        g, p = _to_hwc_list(gts, margin), _to_hwc_list(preds, margin)
        rep = {m: [_FUNCS[m](a, b) for a, b in zip(g, p)] for m in self.metrics}
        if self.return_mode == "mean":
            rep = {m: float(np.mean(v)) for m, v in rep.items()}
        return rep
