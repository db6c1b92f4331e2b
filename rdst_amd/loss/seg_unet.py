"""Seg-UNet perceptual loss in the backward path (BASELINE.json configs[4] = RDST-HRL): the
reference's ``SegUNet_F`` (loss/seg_unet.py) with every mode it has — 'encoder*' (feature layers 0..5), 'decoder*',
'label-hr' (RDST-HRL: multiclass Dice against the UNet's own segmentation of the HR image) and 'label-gt'.

What the reference does (loss/seg_unet.py:46-127): ``segmentation_models_pytorch.Unet(in_channels, classes=4)`` — resnet34
encoder, UNet decoder (256, 128, 64, 32, 16) with BatchNorm, 3x3 head — with weights from ``loss/unet_*.pt`` maps the SR
output (with grad) and the HR image (no_grad) to features / decoder output / logits and compares them.  Quirks kept:
a mode name containing 'L1' selects **MSELoss** and 'L2' selects L1Loss (:73-78); ``loss /= len(layers)`` sits INSIDE the
loop (:105-107); the UNet is never put in ``eval()``, so every BatchNorm uses batch statistics and updates its running
statistics on the SR batch and then on the HR batch of every call; ``requires_grad = False`` lands on the modules, not on
the parameters (:59-61) — the reference therefore also computes gradients of the UNet's own weights, which no optimizer
ever reads (models/trans_sr_trainer.py:72 optimises model_g only): they are NOT produced here, the backward goes to the SR
image alone.

**Parity unpinned**: neither segmentation_models_pytorch nor the weight files exist in the build image (SURVEY.md
section 8c).  The modules below carry smp 0.3.x's state-dict keys (``encoder.*``, ``decoder.blocks.N.conv{1,2}.{0,1}.*``,
``tail.0.*`` = smp's ``segmentation_head.0.*``, as the reference renames it at :57), so a reference ``unet_oasis.pt`` loads
strictly; the arithmetic is checked against oracle/segunet_oracle.py (a plain-torch restatement of smp's published
architecture and Dice loss), never against smp itself.

The network runs on the HIP entry points ``rdst_u_*`` (include/rdst_hip.h: implicit-GEMM convolutions on the matrix
cores with the decoder's upsample + concat folded into the loads, training-mode BatchNorm with fixed-order statistics,
Dice / feature losses); torch holds the buffers.  There is no CPU fallback.  ``set_compute_dtype(torch.bfloat16)`` selects
the throughput mode (bf16 activations, fp32 accumulation and statistics), the default fp32 is the parity mode.
"""
from __future__ import annotations

import os
import ctypes
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .._lib import BF16, F32, F32X3

RESNET34_BLOCKS = (3, 4, 6, 3)
RESNET34_PLANES = (64, 128, 256, 512)
DECODER_CHANNELS = (256, 128, 64, 32, 16)
NCLS_PAD = 16     # the Dice gradient is written 16 channels wide (zeros beyond `classes`): a 32-byte reduction for the head's dgrad


# ------------------------------------------------------------------------------------------------------------------------
# parameter containers under smp's state-dict names (they never run: the runner below reads their tensors)
# ------------------------------------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


class _Encoder(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, (nb, pl) in enumerate(zip(RESNET34_BLOCKS, RESNET34_PLANES), start=1):
            blocks = []
            for b in range(nb):
                blocks.append(_BasicBlock(inpl, pl, 2 if (b == 0 and li > 1) else 1))
                inpl = pl
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        for m in self.modules():    # torchvision ResNet.__init__
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(cin + cskip, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        rev = [512, 256, 128, 64, 64]
        in_ch = [rev[0]] + list(DECODER_CHANNELS[:-1])
        skip_ch = rev[1:] + [0]
        self.blocks = nn.ModuleList([_DecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, DECODER_CHANNELS)])
        for m in self.modules():    # smp.base.initialization.initialize_decoder
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")


# ------------------------------------------------------------------------------------------------------------------------
# the UNet on the rdst_u_* entry points
# ------------------------------------------------------------------------------------------------------------------------
def _ptr(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    """Row stride (elements) of a (B, H, W, C) activation or a channel slice of one."""
    if t is None:
        return 0
    B, H, W, C = t.shape
    ld = t.stride(2)
    if t.stride(3) != 1 or (H > 1 and t.stride(1) != W * ld) or (B > 1 and t.stride(0) != H * W * ld):
        raise ValueError("rdst_amd.loss: activation is not a row view")
    return ld


class _Runner:
    """One forward (+ backward) of the UNet for one module state; all launches on torch's current stream."""

    def __init__(self, mod: "SegUNet_F"):
        self.m = mod
        self.lib = _lib.load()
        self.dt = mod.compute_dtype
        self.code = mod.compute_code
        self.dev = mod.encoder.conv1.weight.device
        self.scratch = mod._scratch()
        self.packs = mod._packs()
        self.st = torch.cuda.current_stream().cuda_stream
        # fp32 / fp32x3: relu(BatchNorm(.)) in front of a 3x3 stride-1 convolution is applied while that convolution stages its
        # input (rdst_u_conv's bn1) and its backward recomputes the ReLU mask from the raw tensor (rdst_u_bn_bwd with
        # mask = raw): the activation tensors a1 of every block and the outputs of decoder blocks 0..3 are never written
        self.fuse = self.code in (_lib.F32, _lib.F32X3)

    # ---- thin wrappers -------------------------------------------------------------------------------------------------
    def conv(self, x1, name, *, transposed=False, x2=None, up1=False, stride=1, out_hw=None, add=None, bias=None, cout=None,
             bn1=None, bn=None):
        """bn1 = the BatchNorm coefficients of x1 when x1 is a RAW convolution output: the convolution then reads it through
        relu(BatchNorm(x1)) (see `fuse`), which is never written.  bn = the nn.BatchNorm2d behind this convolution: returns
        (y, coef) with the statistics taken from the convolution's own epilogue where the kernel offers that (3x3 stride 1),
        from a pass over y otherwise."""
        wp, npad, k = self.packs[(name, transposed)]
        B, H1, W1, C1 = x1.shape
        Hin, Win = (2 * H1, 2 * W1) if up1 else (H1, W1)
        C2 = 0 if x2 is None else x2.shape[-1]
        if out_hw is None:
            out_hw = ((Hin - 1) // stride + 1, (Win - 1) // stride + 1) if not transposed else (Hin * stride, Win * stride)
        y = torch.empty((B, out_hw[0], out_hw[1], cout), dtype=self.dt, device=self.dev)
        epi = (bn is not None and self.fuse and k == 3 and stride == 1 and not transposed and bias is None and add is None
               and (C1 * 4) % 64 == 0 and ((C1 + C2) * 4) % 64 == 0)
        if epi:
            part = torch.empty((B * ((out_hw[0] + 7) // 8) * ((out_hw[1] + 7) // 8), 2, cout), dtype=torch.float32, device=self.dev)
            nblk = ctypes.c_int(0)
        _lib.check(self.lib.rdst_u_conv(x1.data_ptr(), _ld(x1), C1, int(up1), _ptr(x2), _ld(x2), C2, wp.data_ptr(), _ptr(bias),
                                        _ptr(add), _ld(add), y.data_ptr(), cout, B, Hin, Win, out_hw[0], out_hw[1], cout, npad, k,
                                        stride, int(transposed), self.code, self.st, _ptr(bn1), part.data_ptr() if epi else None,
                                        ctypes.byref(nblk) if epi else None), "rdst_u_conv")
        if bn is None:
            return y
        if not epi:
            return y, self.bn_stats(y, bn)
        coef = torch.empty(4 * cout, dtype=torch.float32, device=self.dev)
        mom = 0.1 if bn.momentum is None else bn.momentum
        _lib.check(self.lib.rdst_u_bn_stats_from(part.data_ptr(), nblk.value, y.numel() // cout, cout, bn.weight.data_ptr(),
                                                 bn.bias.data_ptr(), float(bn.eps), float(mom), _ptr(bn.running_mean),
                                                 _ptr(bn.running_var), coef.data_ptr(), self.scratch.data_ptr(), self.st),
                   "rdst_u_bn_stats_from")
        return y, coef

    def bn_stats(self, x, bn: nn.BatchNorm2d, update=True):
        C = x.shape[-1]
        coef = torch.empty(4 * C, dtype=torch.float32, device=self.dev)
        P = x.numel() // C
        mom = 0.1 if bn.momentum is None else bn.momentum
        _lib.check(self.lib.rdst_u_bn_stats(x.data_ptr(), _ld(x), P, C, bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(mom),
                                            _ptr(bn.running_mean) if update else None, _ptr(bn.running_var) if update else None,
                                            coef.data_ptr(), self.scratch.data_ptr(), self.code, self.st), "rdst_u_bn_stats")
        return coef

    def bn_apply(self, x, coef, relu=True, x2=None, coef2=None, res=None):
        y = torch.empty(x.shape, dtype=self.dt, device=self.dev)
        C = x.shape[-1]
        _lib.check(self.lib.rdst_u_bn_apply(x.data_ptr(), _ld(x), coef.data_ptr(), _ptr(x2), _ld(x2), _ptr(coef2), _ptr(res), _ld(res),
                                            int(relu), y.data_ptr(), C, x.numel() // C, C, self.code, self.st), "rdst_u_bn_apply")
        return y

    def bn_bwd(self, dy, mask, raw, coef, want_g=False, gadd=None):
        C = raw.shape[-1]
        dx = torch.empty(raw.shape, dtype=self.dt, device=self.dev)
        g = torch.empty(raw.shape, dtype=self.dt, device=self.dev) if want_g else None
        _lib.check(self.lib.rdst_u_bn_bwd(dy.data_ptr(), _ld(dy), _ptr(mask), _ld(mask), raw.data_ptr(), _ld(raw), coef.data_ptr(),
                                          dx.data_ptr(), C, _ptr(g), C if want_g else 0, _ptr(gadd), _ld(gadd), raw.numel() // C, C,
                                          self.scratch.data_ptr(), self.code, self.st), "rdst_u_bn_bwd")
        return (dx, g) if want_g else dx

    # ---- forward ------------------------------------------------------------------------------------------------------
    def encoder(self, img, depth, save: Optional[dict]):
        """features[1..depth] (features[0] is the image itself).  `save` collects what the backward needs."""
        enc = self.m.encoder
        B, Cin, H, W = img.shape
        img = img.detach().float().contiguous()
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        raw = torch.empty((B, Ho, Wo, 64), dtype=self.dt, device=self.dev)
        _lib.check(self.lib.rdst_u_stem_fwd(img.data_ptr(), enc.conv1.weight.data_ptr(), raw.data_ptr(), 64, B, Cin, H, W, self.code,
                                            self.st), "rdst_u_stem_fwd")
        coef = self.bn_stats(raw, enc.bn1)
        f1 = self.bn_apply(raw, coef)
        feats = [None, f1]
        if save is not None:
            save["stem"] = (raw, coef, f1, (B, Cin, H, W))
            save["depth"] = depth
        if depth < 2:
            return feats
        Hp, Wp = (Ho + 1) // 2, (Wo + 1) // 2
        pool = torch.empty((B, Hp, Wp, 64), dtype=self.dt, device=self.dev)
        idx = torch.empty((B, Hp, Wp, 64), dtype=torch.uint8, device=self.dev)
        _lib.check(self.lib.rdst_u_maxpool_fwd(f1.data_ptr(), 64, pool.data_ptr(), 64, idx.data_ptr(), B, Ho, Wo, 64, self.code, self.st),
                   "rdst_u_maxpool_fwd")
        if save is not None:
            save["pool"] = (idx, (B, Ho, Wo))
            save["blocks"] = []
        x = pool
        for li in range(1, depth):
            layer = getattr(enc, f"layer{li}")
            for bi, blk in enumerate(layer):
                name = f"encoder.layer{li}.{bi}"
                pl = blk.conv1.out_channels
                r1, c1 = self.conv(x, name + ".conv1", stride=blk.stride, cout=pl, bn=blk.bn1)
                if self.fuse:
                    a1 = None
                    r2, c2 = self.conv(r1, name + ".conv2", cout=pl, bn1=c1, bn=blk.bn2)
                else:
                    a1 = self.bn_apply(r1, c1)
                    r2, c2 = self.conv(a1, name + ".conv2", cout=pl, bn=blk.bn2)
                if blk.downsample is not None:
                    rd = self.conv(x, name + ".downsample.0", stride=blk.stride, cout=pl)
                    cd = self.bn_stats(rd, blk.downsample[1])
                    out = self.bn_apply(r2, c2, x2=rd, coef2=cd)
                else:
                    rd = cd = None
                    out = self.bn_apply(r2, c2, res=x)
                if save is not None:
                    save["blocks"].append((name, blk.stride, x.shape, r1, c1, a1, r2, c2, rd, cd, out))
                x = out
            feats.append(x)
        return feats

    def decoder(self, feats, save: Optional[dict]):
        x = feats[5]
        skips = [feats[4], feats[3], feats[2], feats[1], None]
        if save is not None:
            save["dec"] = []
        nblk = len(self.m.decoder.blocks)
        xcoef = None   # x is a raw convolution output to be read through relu(BatchNorm(x)) with these coefficients
        for i, blk in enumerate(self.m.decoder.blocks):
            name = f"decoder.blocks.{i}"
            co = blk.conv1[0].out_channels
            cx, cs = x.shape[-1], 0 if skips[i] is None else skips[i].shape[-1]
            r1, c1 = self.conv(x, name + ".conv1.0", x2=skips[i], up1=True, cout=co, bn1=xcoef, bn=blk.conv1[1])
            if self.fuse:
                a1 = None
                r2, c2 = self.conv(r1, name + ".conv2.0", cout=co, bn1=c1, bn=blk.conv2[1])
            else:
                a1 = self.bn_apply(r1, c1)
                r2, c2 = self.conv(a1, name + ".conv2.0", cout=co, bn=blk.conv2[1])
            if self.fuse and i + 1 < nblk:   # the last block's output is a feature the losses / the head read: it is written
                out, x, xcoef = None, r2, c2
            else:
                out = self.bn_apply(r2, c2)
                x, xcoef = out, None
            if save is not None:
                save["dec"].append((name, cx, cs, r1, c1, a1, r2, c2, out))
        return x

    def head(self, dec):
        conv = self.m.tail[0]
        return self.conv(dec, "tail.0", bias=conv.bias, cout=conv.out_channels)

    # ---- backward (to the image) ----------------------------------------------------------------------------------------
    def backward(self, save, d_feats: Dict[int, torch.Tensor], d_dec=None, upstream=None):
        """d_feats[k] = gradient w.r.t. features[k] (k = 1..5) from the feature losses; d_dec = gradient w.r.t. the decoder
        output.  Returns d(image) as fp32 NCHW, times the device scalar `upstream`."""
        d_feats = dict(d_feats)
        if d_dec is not None:
            dy = d_dec
            for i in range(len(save["dec"]) - 1, -1, -1):
                name, cx, cs, r1, c1, a1, r2, c2, out = save["dec"][i]
                # (an activation that was never written: mask = the raw tensor itself, recomputed by the kernel)
                dr2 = self.bn_bwd(dy, out if out is not None else r2, r2, c2)
                da1 = self.conv(dr2, name + ".conv2.0", transposed=True, cout=r1.shape[-1])
                dr1 = self.bn_bwd(da1, a1 if a1 is not None else r1, r1, c1)
                dcat = self.conv(dr1, name + ".conv1.0", transposed=True, cout=cx + cs)
                B, H, W, _ = dcat.shape
                k = 5 - i                                    # the feature this block's x came from (block 0: features[5])
                dprev = torch.empty((B, H // 2, W // 2, cx), dtype=self.dt, device=self.dev)
                add = d_feats.pop(5, None) if i == 0 else None
                _lib.check(self.lib.rdst_u_sumpool2(dcat.data_ptr(), cx + cs, _ptr(add), _ld(add), dprev.data_ptr(), cx, B, H // 2, W // 2,
                                                    cx, self.code, self.st), "rdst_u_sumpool2")
                if cs:
                    dskip = dcat[..., cx:]
                    kk = k - 1                               # skips: features[4], [3], [2], [1]
                    if kk in d_feats:
                        dskip = _add_rows(d_feats[kk], dskip)
                    d_feats[kk] = dskip
                dy = dprev
            d_feats[5] = dy
        # encoder, deepest layer first.  Layer li maps features[li] (layer 1: the max-pooled features[1]) to features[li + 1];
        # `dy` = gradient w.r.t. the output of the block at hand, None while no loss reaches this depth.
        blocks = save.get("blocks", [])
        depth = save["depth"]
        dy = d_feats.pop(depth, None) if depth >= 2 else None
        for li in range(depth - 1, 0, -1):
            layer = [b for b in blocks if b[0].startswith(f"encoder.layer{li}.")]
            extra = d_feats.pop(li, None) if li >= 2 else None       # more gradient for this layer's input, features[li]
            if dy is None:
                dy = extra
                continue
            for name, stride, xshape, r1, c1, a1, r2, c2, rd, cd, out in reversed(layer):
                first = name.endswith(".0")
                cin = xshape[-1]
                if rd is None:
                    if first and extra is not None:
                        raise RuntimeError("SegUNet_F: a layer >= 2 without a downsample branch")   # not a resnet34
                    dr2, g = self.bn_bwd(dy, out, r2, c2, want_g=True)
                    da1 = self.conv(dr2, name + ".conv2", transposed=True, cout=r1.shape[-1])
                    dr1 = self.bn_bwd(da1, a1 if a1 is not None else r1, r1, c1)
                    dy = self.conv(dr1, name + ".conv1", transposed=True, cout=cin, add=g)
                else:
                    dr2 = self.bn_bwd(dy, out, r2, c2)
                    drd = self.bn_bwd(dy, out, rd, cd)
                    da1 = self.conv(dr2, name + ".conv2", transposed=True, cout=r1.shape[-1])
                    dr1 = self.bn_bwd(da1, a1 if a1 is not None else r1, r1, c1)
                    hw = (xshape[1], xshape[2])
                    tmp = self.conv(drd, name + ".downsample.0", transposed=True, stride=stride, out_hw=hw, cout=cin,
                                    add=extra if first else None)
                    dy = self.conv(dr1, name + ".conv1", transposed=True, stride=stride, out_hw=hw, cout=cin, add=tmp)
        # max pool + stem
        raw, coef, f1, (B, Cin, H, W) = save["stem"]
        df1 = d_feats.pop(1, None)
        if depth >= 2 and dy is not None:
            idx, (Bp, Ho, Wo) = save["pool"]
            t = torch.empty(f1.shape, dtype=self.dt, device=self.dev)
            _lib.check(self.lib.rdst_u_maxpool_bwd(dy.data_ptr(), _ld(dy), idx.data_ptr(), _ptr(df1), _ld(df1), t.data_ptr(), 64, Bp, Ho, Wo,
                                                   64, self.code, self.st), "rdst_u_maxpool_bwd")
            df1 = t
        dimg = torch.empty((B, Cin, H, W), dtype=torch.float32, device=self.dev)
        if df1 is None:
            return dimg.zero_()
        draw = self.bn_bwd(df1, f1, raw, coef)
        _lib.check(self.lib.rdst_u_stem_dgrad(draw.data_ptr(), 64, self.m.encoder.conv1.weight.data_ptr(), _ptr(upstream), dimg.data_ptr(),
                                              B, Cin, H, W, self.code, self.st), "rdst_u_stem_dgrad")
        return dimg


def _pack_weights(t, code):
    """(taps, Npad, K) fp32 -> the image rdst_u_conv reads (include/rdst_hip.h): bf16 / fp32 as they are; RDST_F32X3 as
    64-byte groups [16 bf16 hi][16 bf16 lo] per 16 reduction elements (K % 16 == 0)."""
    if code == BF16:
        return t.to(torch.bfloat16).contiguous()
    if code == F32:
        return t.float().contiguous()
    taps, n, K = t.shape
    if K % 16:
        raise ValueError("rdst_amd.loss: the fp32x3 mode needs a reduction length that is a multiple of 16")
    hi = t.float().to(torch.bfloat16)
    lo = (t.float() - hi.float()).to(torch.bfloat16)
    return torch.cat([hi.reshape(taps, n, K // 16, 16), lo.reshape(taps, n, K // 16, 16)], dim=-1).reshape(taps, n, 2 * K).contiguous()


def _add_rows(a, b):
    """a + b for two row views (rare: a feature that receives both a feature-loss gradient and a skip gradient)."""
    return (a.float() + b.float()).to(a.dtype)


class _UNetLoss(torch.autograd.Function):
    """loss = SegUNet_F's value for the 'encoder' (any layers), 'decoder', 'label-hr' and 'label-gt' modes; gradient to `sr`."""

    @staticmethod
    def forward(ctx, sr, hr, gt_label, mod):
        if not sr.is_cuda:
            raise RuntimeError("rdst_amd.loss.SegUNet_F: the UNet runs on HIP kernels; there is no CPU fallback")
        r = _Runner(mod)
        lib, code, st = r.lib, r.code, r.st
        mode = mod.loss_mode
        loss = torch.zeros((), dtype=torch.float32, device=sr.device)
        save: dict = {}
        ctx.mod, ctx.save, ctx.kind = mod, save, None
        if "encoder" in mode:
            layers = mod.loss_layers
            # The reference's encoder always runs all five stages (loss/seg_unet.py:84: `self.encoder(x)`), whatever layers
            # the loss reads.  The stages beyond max(layers) feed nothing but the running statistics of their own
            # BatchNorms, which nothing ever reads (the UNet is never put in eval(): every pass normalises with batch
            # statistics).  By default they are skipped — those BatchNorms' buffers (running_mean / running_var /
            # num_batches_tracked under checkpoint['loss']) then stay where they were, a stated divergence from the
            # reference's saved state; `mod.full_encoder_stats = True` runs them for a bit-faithful state at their cost.
            depth = 5 if mod.full_encoder_stats else max([l for l in layers] + [1])
            fs = r.encoder(sr, depth, save)
            fh = r.encoder(hr, depth, None)
            n = len(layers)
            terms = []
            for i, l in enumerate(layers):
                w = float(n) ** -(n - i)                     # loss/seg_unet.py:105-107: `loss /= len(layers)` inside the loop
                if l == 0:
                    terms.append((0, w))
                    continue
                a, b = fs[l], fh[l]
                C = a.shape[-1]
                _lib.check(lib.rdst_u_pair_loss_fwd(a.data_ptr(), C, b.data_ptr(), C, a.numel() // C, C, int(mod.use_mse), w, 1,
                                                    loss.data_ptr(), r.scratch.data_ptr(), code, st), "rdst_u_pair_loss_fwd")
                terms.append((l, w))
            save["pairs"] = (fs, fh, terms)
            if any(l == 0 for l, _ in terms):                # features[0] is the image itself (encoders/resnet.py: nn.Identity())
                w0 = sum(w for l, w in terms if l == 0)
                d = sr.detach().float() - hr.detach().float()
                loss = loss + w0 * (d.pow(2).mean() if mod.use_mse else d.abs().mean())
                save["img"] = (sr.detach(), hr.detach(), w0)
            ctx.kind = "encoder"
        else:
            fs = r.encoder(sr, 5, save)
            dec_s = r.decoder(fs, save)
            if mode != "label-gt":                   # 'label-gt' never looks at HR (loss/seg_unet.py:117-123)
                fh = r.encoder(hr, 5, None)
                dec_h = r.decoder(fh, None)
            if "decoder" in mode:
                C = dec_s.shape[-1]
                _lib.check(lib.rdst_u_pair_loss_fwd(dec_s.data_ptr(), C, dec_h.data_ptr(), C, dec_s.numel() // C, C, int(mod.use_mse), 1.0,
                                                    0, loss.data_ptr(), r.scratch.data_ptr(), code, st), "rdst_u_pair_loss_fwd")
                save["decpair"] = (dec_s, dec_h)
                ctx.kind = "decoder"
            else:
                lg_s = r.head(dec_s)
                ncls = lg_s.shape[-1]
                P = lg_s.numel() // ncls
                if mode == "label-hr":
                    lg_h, labels = r.head(dec_h), None
                elif mode == "label-gt":
                    if gt_label is None:
                        raise ValueError("SegUNet_F('label-gt') needs gt_label")
                    labels = gt_label[:, 0] if gt_label.dim() == 4 else gt_label     # :118-120
                    labels = labels.to(torch.long).contiguous()
                    lg_h = None
                else:
                    raise ValueError("Invalid UNet Seg Loss Mode: {}".format(mode))
                coef = torch.empty(16, dtype=torch.float32, device=sr.device)
                cmask = sum(1 << c for c in mod.dice_classes)
                _lib.check(lib.rdst_u_dice_fwd(lg_s.data_ptr(), ncls, _ptr(lg_h), ncls, _ptr(labels), P, ncls, cmask, 1e-7, 1.0, 0,
                                               loss.data_ptr(), coef.data_ptr(), r.scratch.data_ptr(), code, st), "rdst_u_dice_fwd")
                save["dice"] = (lg_s, lg_h, labels, coef, dec_s.shape)
                if mod.keep_debug:
                    mod.debug_last = {"sr_logits": lg_s, "hr_logits": lg_h}
                ctx.kind = "label"
        # only the BatchNorms that ran count a batch (nn.BatchNorm2d.forward in training mode): the encoder stages up to
        # `depth` twice in the 'encoder' modes and never the decoder's; everything once ('label-gt') or twice otherwise
        mod._count_batches(1 if mode == "label-gt" else 2, depth if "encoder" in mode else None)
        return loss

    @staticmethod
    def backward(ctx, gout):
        mod, save = ctx.mod, ctx.save
        r = _Runner(mod)
        lib, code, st = r.lib, r.code, r.st
        up = gout.detach().float().contiguous()
        d_feats, d_dec = {}, None
        if ctx.kind == "encoder":
            fs, fh, terms = save["pairs"]
            for l, w in terms:
                if l == 0:
                    continue
                a, b = fs[l], fh[l]
                C = a.shape[-1]
                prev = d_feats.get(l)
                g = torch.empty(a.shape, dtype=a.dtype, device=a.device)
                _lib.check(lib.rdst_u_pair_loss_bwd(a.data_ptr(), C, b.data_ptr(), C, a.numel() // C, C, int(mod.use_mse), w, None,
                                                    _ptr(prev), _ld(prev), g.data_ptr(), C, code, st), "rdst_u_pair_loss_bwd")
                d_feats[l] = g
        elif ctx.kind == "decoder":
            a, b = save["decpair"]
            C = a.shape[-1]
            d_dec = torch.empty(a.shape, dtype=a.dtype, device=a.device)
            _lib.check(lib.rdst_u_pair_loss_bwd(a.data_ptr(), C, b.data_ptr(), C, a.numel() // C, C, int(mod.use_mse), 1.0, None, None, 0,
                                                d_dec.data_ptr(), C, code, st), "rdst_u_pair_loss_bwd")
        else:
            lg_s, lg_h, labels, coef, dshape = save["dice"]
            ncls = lg_s.shape[-1]
            P = lg_s.numel() // ncls
            dlg = torch.empty(lg_s.shape[:-1] + (NCLS_PAD,), dtype=lg_s.dtype, device=lg_s.device)
            _lib.check(lib.rdst_u_dice_bwd(lg_s.data_ptr(), ncls, _ptr(lg_h), ncls, _ptr(labels), P, ncls, coef.data_ptr(), None,
                                           dlg.data_ptr(), NCLS_PAD, NCLS_PAD, code, st), "rdst_u_dice_bwd")
            d_dec = r.conv(dlg, "tail.0", transposed=True, cout=dshape[-1])
        dimg = r.backward(save, d_feats, d_dec, upstream=up)
        if "img" in save:
            s, h, w0 = save["img"]
            d = s.float() - h.float()
            gi = (2.0 * d if mod.use_mse else torch.sign(d)) * (w0 / d.numel())
            dimg = dimg + gi * up
        ctx.save = None
        return dimg, None, None, None


class SegUNet_F(nn.Module):
    _MODES = {"OASIS": (1, "loss/unet_oasis.pt"), "BraTS": (4, "loss/unet_brats.pt"), "ACDC": (1, "loss/unet_acdc.pt"),
              "COVID": (1, "loss/unet_covid.pt")}

    def __init__(self, loss_layers: Dict[str, list], mode: str = "OASIS", unet_path: Optional[str] = None,
                 allow_random_init: bool = False, classes: int = 4):
        super().__init__()
        in_channels = None
        for k, (c, path) in self._MODES.items():
            if k in mode:
                in_channels, default_path = c, path
        if in_channels is None:
            raise ValueError("Invalid UNet Seg Loss data mode: {}".format(mode))
        self.dice_classes = [0, 1, 2, 3]
        if "tumor_only" in mode or "lesion_only" in mode:       # loss/seg_unet.py:40-44
            self.dice_classes = [1, 2, 3]
        for k in loss_layers:                         # loss/seg_unet.py:51-53: the last key wins
            self.loss_mode = k
        self.loss_layers = list(loss_layers[self.loss_mode])
        if not any(s in self.loss_mode for s in ("encoder", "decoder")) and self.loss_mode not in ("label-hr", "label-gt"):
            raise ValueError("Invalid UNet Seg Loss Mode: {}".format(self.loss_mode))
        if "encoder" in self.loss_mode and any(l not in range(6) for l in self.loss_layers):
            raise ValueError("SegUNet_F: encoder loss layers are 0..5 (the six feature maps of the resnet34 encoder)")
        self.encoder = _Encoder(in_channels)
        self.decoder = _Decoder()
        self.tail = nn.Sequential(nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1))
        nn.init.xavier_uniform_(self.tail[0].weight)             # smp.base.initialization.initialize_head
        nn.init.constant_(self.tail[0].bias, 0)
        path = unet_path if unet_path is not None else default_path
        if os.path.exists(path):                      # a reference UNet checkpoint (smp.Unet.state_dict())
            sd = torch.load(path, map_location="cpu")
            sd = {("tail." + k[len("segmentation_head."):] if k.startswith("segmentation_head.") else k): v for k, v in sd.items()}
            self.load_state_dict(sd, strict=True)
        elif not allow_random_init:
            raise ValueError("Pre-trained UNet not exist: {}".format(path))       # loss/seg_unet.py:47-48
        for p in self.parameters():                   # the reference's intent (:59-61); nothing here computes their gradients
            p.requires_grad_(False)
        self.loss_names = ["SegUNet({})".format(self.loss_mode)]
        self.use_mse = "L1" in self.loss_mode         # loss/seg_unet.py:73-78: 'L1' -> MSELoss, everything else L1Loss
        self.compute_dtype, self.compute_code = torch.float32, F32X3
        self.full_encoder_stats = False               # see _UNetLoss.forward ('encoder' modes)
        self.keep_debug = False                       # tests: keep the logits of the last 'label' call in self.debug_last
        self._pack_cache = {}
        self._scratch_buf = None

    # ---- plumbing ----------------------------------------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        """'fp32x3' (default): fp32 activations, convolutions as a 3-term bf16 split on the matrix cores (~1e-5 relative);
        'fp32' / torch.float32: exact fp32 MFMA (the parity mode, 5x slower GEMMs); 'bf16' / torch.bfloat16: bf16 activations
        (fastest, but the loss DIFFERENCES SR and HR features that then carry 1e-2..1e-1 relative error: noisy gradients)."""
        table = {"fp32x3": (torch.float32, F32X3), "fp32": (torch.float32, F32), torch.float32: (torch.float32, F32),
                 "bf16": (torch.bfloat16, BF16), torch.bfloat16: (torch.bfloat16, BF16)}
        if dtype not in table:
            raise ValueError("compute dtype must be 'fp32x3', 'fp32' / torch.float32 or 'bf16' / torch.bfloat16")
        self.compute_dtype, self.compute_code = table[dtype]
        return self

    def _scratch(self):
        dev = self.encoder.conv1.weight.device
        if self._scratch_buf is None or self._scratch_buf.device != dev:
            self._scratch_buf = torch.empty(_lib.load().rdst_u_scratch_bytes(), dtype=torch.uint8, device=dev)
        return self._scratch_buf

    def _conv_weights(self):
        for n, m in self.named_modules():
            if isinstance(m, nn.Conv2d) and n != "encoder.conv1":
                yield n, m.weight

    def _packs(self):
        """name, transposed -> (weights as [k*k][Npad][K] in the compute dtype, Npad, k): frozen weights, packed once per
        (weights version, dtype, device)."""
        ws = list(self._conv_weights())
        sig = (self.compute_code, tuple((w.data_ptr(), w._version) for _, w in ws))
        if self._pack_cache.get("sig") != sig:
            packs = {}
            with torch.no_grad():
                for n, w in ws:
                    co, ci, k, _ = w.shape
                    for tr in (False, True):
                        t = w.permute(2, 3, 1, 0) if tr else w.permute(2, 3, 0, 1)       # (k, k, N, K)
                        N, K = t.shape[2], t.shape[3]
                        npad = (N + 31) // 32 * 32
                        kpad = max(K, NCLS_PAD) if tr and n == "tail.0" else K           # the head's dgrad reduces over 16 padded classes
                        t = F.pad(t, (0, kpad - K, 0, npad - N)).reshape(k * k, npad, kpad)
                        packs[(n, tr)] = (_pack_weights(t, self.compute_code), npad, k)
            self._pack_cache = {"sig": sig, "packs": packs}
        return self._pack_cache["packs"]

    def _bns_up_to(self, depth):
        """The BatchNorms `_Runner.encoder(img, depth)` runs: the stem's and those of layer1 .. layer{depth-1}."""
        ran = {id(self.encoder.bn1)}
        for li in range(1, depth):
            ran.update(id(m) for m in getattr(self.encoder, f"layer{li}").modules() if isinstance(m, nn.BatchNorm2d))
        return ran

    def _count_batches(self, n, encoder_depth=None):
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        flat = getattr(self, "_nbt_flat", None)
        ok = flat is not None and flat.device == bns[0].num_batches_tracked.device and all(
            m.num_batches_tracked.data_ptr() == flat.data_ptr() + 8 * i for i, m in enumerate(bns))
        if not ok:        # one flat counter buffer behind every num_batches_tracked: one add per call instead of one per layer
            flat = torch.stack([m.num_batches_tracked.reshape(()) for m in bns]).contiguous()
            for i, m in enumerate(bns):
                m.num_batches_tracked = flat[i]
            object.__setattr__(self, "_nbt_flat", flat)
            object.__setattr__(self, "_nbt_masks", {})
        if encoder_depth is None:
            flat += n
            return
        mask = self._nbt_masks.get(encoder_depth)
        if mask is None or mask.device != flat.device:
            ran = self._bns_up_to(encoder_depth)
            mask = torch.tensor([1 if id(m) in ran else 0 for m in bns], dtype=flat.dtype, device=flat.device)
            self._nbt_masks[encoder_depth] = mask
        flat.add_(mask, alpha=n)

    # ---- loss/seg_unet.py:94-127 ----------------------------------------------------------------------------------------
    def forward(self, sr, hr, gt_label=None):
        assert sr.shape == hr.shape, "Seg UNet Loss invalid SR({}) and HR({}) shape!".format(sr.shape, hr.shape)
        from .sr_loss import LazyScalars
        loss = _UNetLoss.apply(sr, hr, gt_label, self)
        return loss, LazyScalars({self.loss_names[0]: loss.detach()})
