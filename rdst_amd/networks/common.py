"""Conv-side building blocks of the RDST hot path on HIP kernels.

Counterpart of the reference's ``networks/common.py`` for the three symbols RDSTSR uses
(``default_conv`` :6-9, ``UpSampler`` :125-148, ``MeanShift`` :151-167): same constructor
signatures, same parameter names/shapes (so state dicts interchange), forward on token-major rows
through ``rdst_amd.ops.conv_rows`` (K4/K5/K6 of include/rdst_hip.h).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import ops


class Conv2d(nn.Conv2d):
    """nn.Conv2d (stride 1, padding k//2, k in {1,3}) whose forward runs the HIP conv.

    ``forward`` keeps the nn.Conv2d contract (fp32 NCHW in/out); ``forward_rows`` is what the fused
    network uses: token-major (B,H,W,Cin) in/out with the activation/scale/residual/pixel-shuffle
    epilogues of K4/K5."""

    def forward_rows(self, x, *, in_act=ops.ACT_NONE, residual=None, out_scale=1.0, shuffle=1, out_slot=None):
        k = self.kernel_size[0]
        if (self.kernel_size[0] != self.kernel_size[1] or k not in (1, 3) or self.stride != (1, 1)
                or self.padding != (k // 2, k // 2) or self.dilation != (1, 1) or self.groups != 1):
            raise NotImplementedError("rdst_amd Conv2d: only k in {1,3}, stride 1, padding k//2, groups 1")
        return ops.conv_rows(x, self.weight, self.bias, in_act=in_act, residual=residual, out_scale=out_scale,
                             shuffle=shuffle, out_slot=out_slot)

    def forward(self, x):
        return ops.rows_to_nchw(self.forward_rows(ops.nchw_to_rows(x, torch.float32)))


def default_conv(in_channels, out_channels, kernel_size, bias=True):
    return Conv2d(in_channels, out_channels, kernel_size, padding=(kernel_size // 2), bias=bias)


class UpSampler(nn.Sequential):
    """[conv(n, 4n, 3), PixelShuffle(2)] x log2(scale)  (or conv(n, 9n, 3) + PixelShuffle(3)).
    The shuffle is folded into the conv's store (K5)."""

    def __init__(self, conv, scale, n_feats, bn=False, act=None, bias=True):
        if bn:
            raise NotImplementedError("rdst_amd UpSampler: BatchNorm is not on the RDST path (bn_in_conv = None)")
        m = []
        if (scale & (scale - 1)) == 0:
            for _ in range(int(math.log(scale, 2))):
                m.append(conv(n_feats, 4 * n_feats, 3, bias))
                m.append(nn.PixelShuffle(2))
                if act is not None:
                    m.append(act)
        elif scale == 3:
            m.append(conv(n_feats, 9 * n_feats, 3, bias))
            m.append(nn.PixelShuffle(3))
            if act is not None:
                m.append(act)
        else:
            raise NotImplementedError('SR scale {} is not valid.'.format(scale))
        super().__init__(*m)

    def forward_rows(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.PixelShuffle):
                x = m.forward_rows(x, shuffle=mods[i + 1].upscale_factor)
                i += 2
            elif isinstance(m, Conv2d):
                x = m.forward_rows(x)
                i += 1
            else:
                raise NotImplementedError(f"rdst_amd UpSampler: {type(m).__name__} inside the upsampler")
        return x

    def forward(self, x):
        return ops.rows_to_nchw(self.forward_rows(ops.nchw_to_rows(x, torch.float32)))


class MeanShift(Conv2d):
    """Frozen 1x1 conv (x - mean)/std ('sub') or x*std + mean ('add')."""

    def __init__(self, mean=(0.,), std=(1.0,), mode='sub'):
        if len(mean) != len(std):
            raise ValueError('Size of means and stds should be the same')
        nc = len(mean)
        super().__init__(nc, nc, kernel_size=1)
        std = torch.Tensor(std)
        mean = torch.Tensor(mean)
        if mode == 'sub':
            self.weight.data = torch.eye(nc).view(nc, nc, 1, 1) / std.view(nc, 1, 1, 1)
            self.bias.data = -1 * mean / std
        elif mode == 'add':
            self.weight.data = torch.eye(nc).view(nc, nc, 1, 1) * std.view(nc, 1, 1, 1)
            self.bias.data = 1 * mean
        for p in self.parameters():
            p.requires_grad = False
