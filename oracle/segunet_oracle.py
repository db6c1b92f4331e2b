"""CPU oracle for the seg-UNet perceptual loss (BASELINE.json configs[4]) — TEST
INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/`` may import it.

**PARITY UNPINNED.**  The reference's loss (loss/seg_unet.py) gets its arithmetic from a third-party dependency that is
absent from /root/reference and from this image: ``segmentation-models-pytorch>=0.3.0`` (requirements.txt:5) —
``smp.Unet(in_channels, classes=4)`` with its defaults (encoder 'resnet34', depth 5, decoder channels (256, 128, 64, 32, 16),
decoder batch-norm on, no attention, no activation; loss/seg_unet.py:46) and ``smp.losses.DiceLoss('multiclass', classes)``
(:71) — plus the weight file loss/unet_oasis.pt (.MISSING_LARGE_BLOBS).  There is nothing to generate golden vectors
from, so this file restates the PUBLISHED algorithm of smp 0.3.x in plain functional torch ops under smp's state-dict
keys, and what it pins is the reference's own call sites: ``SegUNet_F.unet_forward`` / ``forward`` (loss/seg_unet.py:80-127).

Restated from smp 0.3.x (segmentation_models_pytorch/):
  encoders/resnet.py   ResNetEncoder.forward: features = [x, relu(bn1(conv1(x))), layer1(maxpool(.)), layer2, layer3, layer4]
                       (torchvision resnet34 BasicBlocks [3, 4, 6, 3], fc / avgpool deleted)
  decoders/unet/decoder.py   UnetDecoder: drop features[0], reverse, head = deepest, center = Identity (resnet),
                       DecoderBlock: nearest x2 -> cat(skip) -> Conv2dReLU x2 (conv 3x3 no bias -> BatchNorm -> ReLU)
  base/heads.py        SegmentationHead: conv 3x3 (16 -> classes, bias), no upsampling, no activation
  losses/dice.py       DiceLoss(mode='multiclass', classes, log_loss=False, from_logits=True, smooth=0.0, eps=1e-7):
                       p = log_softmax(dim=1).exp(); one-hot target; per class over dims (0, 2):
                       dice = 2 sum(p t) / clamp_min(sum(p + t), eps); loss_c = (1 - dice) * [sum(t) > 0];
                       mean over ``classes``.
BatchNorm runs in TRAINING mode (the reference never calls eval() on the UNet: loss/seg_unet.py:59-61 only sets an
attribute on the modules), so batch statistics are used and the running statistics are updated per call.

State-dict keys: ``encoder.*``, ``decoder.*`` as smp names them, and ``tail.0.{weight,bias}`` for the segmentation head
(the reference stores it as ``self.tail``, loss/seg_unet.py:57).
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]

RESNET34_BLOCKS = (3, 4, 6, 3)
RESNET34_PLANES = (64, 128, 256, 512)
DECODER_CHANNELS = (256, 128, 64, 32, 16)


def unet_layout(in_channels: int = 1, classes: int = 4) -> Dict[str, tuple]:
    """name -> shape of every state-dict entry of SegUNet_F (parameters and BatchNorm buffers), in registration order."""
    out: Dict[str, tuple] = {}

    def bn(prefix, c):
        out[prefix + ".weight"] = (c,)
        out[prefix + ".bias"] = (c,)
        out[prefix + ".running_mean"] = (c,)
        out[prefix + ".running_var"] = (c,)
        out[prefix + ".num_batches_tracked"] = ()

    out["encoder.conv1.weight"] = (64, in_channels, 7, 7)
    bn("encoder.bn1", 64)
    inpl = 64
    for li, (nb, pl) in enumerate(zip(RESNET34_BLOCKS, RESNET34_PLANES), start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"encoder.layer{li}.{b}"
            out[p + ".conv1.weight"] = (pl, inpl, 3, 3)
            bn(p + ".bn1", pl)
            out[p + ".conv2.weight"] = (pl, pl, 3, 3)
            bn(p + ".bn2", pl)
            if stride != 1 or inpl != pl:
                out[p + ".downsample.0.weight"] = (pl, inpl, 1, 1)
                bn(p + ".downsample.1", pl)
            inpl = pl
    enc = [64, 64, 128, 256, 512]                       # features[1:]
    rev = enc[::-1]                                     # 512, 256, 128, 64, 64
    in_ch = [rev[0]] + list(DECODER_CHANNELS[:-1])
    skip_ch = rev[1:] + [0]
    for i, (ic, sc, oc) in enumerate(zip(in_ch, skip_ch, DECODER_CHANNELS)):
        p = f"decoder.blocks.{i}"
        out[p + ".conv1.0.weight"] = (oc, ic + sc, 3, 3)
        bn(p + ".conv1.1", oc)
        out[p + ".conv2.0.weight"] = (oc, oc, 3, 3)
        bn(p + ".conv2.1", oc)
    out["tail.0.weight"] = (classes, DECODER_CHANNELS[-1], 3, 3)
    out["tail.0.bias"] = (classes,)
    return out


def make_unet_weights(in_channels: int = 1, classes: int = 4, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic weights keyed by state-dict name (numpy PCG64), scaled so activations stay O(1) through 34 layers."""
    import zlib
    import numpy as np
    sd = {}
    for k, shp in unet_layout(in_channels, classes).items():
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(k.encode())]))
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shp, dtype=dtype)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp, dtype=dtype)
        elif len(shp) == 4:
            fan = shp[1] * shp[2] * shp[3]
            sd[k] = torch.from_numpy((rng.standard_normal(shp) * (2.0 / fan) ** 0.5).astype("float32")).to(dtype)
        elif k.endswith(".weight"):
            sd[k] = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(shp)).astype("float32")).to(dtype)
        else:
            sd[k] = torch.from_numpy((0.1 * rng.standard_normal(shp)).astype("float32")).to(dtype)
    return sd


class BNState:
    """Running statistics as nn.BatchNorm2d in training mode keeps them (momentum 0.1, unbiased variance)."""

    def __init__(self, sd: SD, update: bool = True):
        self.sd = sd
        self.update = update

    def __call__(self, x: Tensor, prefix: str, eps: float = 1e-5, momentum: float = 0.1) -> Tensor:
        sd = self.sd
        rm, rv = sd.get(prefix + ".running_mean"), sd.get(prefix + ".running_var")
        if self.update and rm is not None:
            y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], True, momentum, eps)
            if prefix + ".num_batches_tracked" in sd:
                sd[prefix + ".num_batches_tracked"] += 1
            return y
        return F.batch_norm(x, None, None, sd[prefix + ".weight"], sd[prefix + ".bias"], True, momentum, eps)


def encoder_forward(x: Tensor, sd: SD, bn: BNState) -> List[Tensor]:
    """smp ResNetEncoder.forward (encoders/resnet.py): the six feature maps."""
    feats = [x]
    y = F.relu(bn(F.conv2d(x, sd["encoder.conv1.weight"], None, 2, 3), "encoder.bn1"))
    feats.append(y)
    y = F.max_pool2d(y, 3, 2, 1)
    inpl = 64
    for li, (nb, pl) in enumerate(zip(RESNET34_BLOCKS, RESNET34_PLANES), start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"encoder.layer{li}.{b}"
            idn = y
            o = F.relu(bn(F.conv2d(y, sd[p + ".conv1.weight"], None, stride, 1), p + ".bn1"))
            o = bn(F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
            if p + ".downsample.0.weight" in sd:
                idn = bn(F.conv2d(y, sd[p + ".downsample.0.weight"], None, stride, 0), p + ".downsample.1")
            y = F.relu(o + idn)
            inpl = pl
        feats.append(y)
    return feats


def decoder_forward(feats: Sequence[Tensor], sd: SD, bn: BNState) -> Tensor:
    """smp UnetDecoder.forward (decoders/unet/decoder.py)."""
    fs = list(feats[1:])[::-1]
    x, skips = fs[0], fs[1:]
    for i in range(len(DECODER_CHANNELS)):
        p = f"decoder.blocks.{i}"
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i < len(skips):
            x = torch.cat([x, skips[i]], dim=1)
        x = F.relu(bn(F.conv2d(x, sd[p + ".conv1.0.weight"], None, 1, 1), p + ".conv1.1"))
        x = F.relu(bn(F.conv2d(x, sd[p + ".conv2.0.weight"], None, 1, 1), p + ".conv2.1"))
    return x


def head_forward(x: Tensor, sd: SD) -> Tensor:
    return F.conv2d(x, sd["tail.0.weight"], sd["tail.0.bias"], 1, 1)


def unet_forward(x: Tensor, sd: SD, loss_mode: str, bn: BNState):
    """loss/seg_unet.py:80-92 (padding_flag is never set: no reflection padding)."""
    feats = encoder_forward(x, sd, bn)
    if "encoder" in loss_mode:
        return feats
    dec = decoder_forward(feats, sd, bn)
    if "decoder" in loss_mode:
        return dec
    return head_forward(dec, sd)


def dice_loss_multiclass(logits: Tensor, target: Tensor, classes: Optional[Sequence[int]] = None, eps: float = 1e-7) -> Tensor:
    """smp.losses.DiceLoss('multiclass', classes) with its defaults (losses/dice.py, losses/_functional.py:soft_dice_score)."""
    bs, nc = logits.shape[0], logits.shape[1]
    p = logits.log_softmax(dim=1).exp().reshape(bs, nc, -1)
    t = F.one_hot(target.reshape(bs, -1), nc).permute(0, 2, 1).to(p.dtype)
    inter = (p * t).sum((0, 2))
    card = (p + t).sum((0, 2))
    dice = 2.0 * inter / card.clamp_min(eps)
    loss = (1.0 - dice) * (t.sum((0, 2)) > 0).to(p.dtype)
    if classes is not None:
        loss = loss[list(classes)]
    return loss.mean()


def pixel_loss(loss_mode: str):
    """loss/seg_unet.py:70-78: a mode name containing 'L1' selects MSELoss, 'L2' (and everything else) L1Loss."""
    return F.mse_loss if "L1" in loss_mode else F.l1_loss


def segunet_loss(sr: Tensor, hr: Tensor, sd: SD, loss_mode: str, loss_layers: Sequence[int], gt_label: Optional[Tensor] = None,
                 dice_classes: Sequence[int] = (0, 1, 2, 3), update_bn: bool = True) -> Tensor:
    """SegUNet_F.forward (loss/seg_unet.py:94-127): SR pass with grad first, then the HR pass under no_grad."""
    assert sr.shape == hr.shape
    bn = BNState(sd, update_bn)
    sr_f = unet_forward(sr, sd, loss_mode, bn)
    if "encoder" in loss_mode:
        with torch.no_grad():
            hr_f = unet_forward(hr, sd, loss_mode, bn)
        lf = pixel_loss(loss_mode)
        loss = 0
        for l in loss_layers:
            loss = loss + lf(sr_f[l], hr_f[l])
            loss = loss / len(loss_layers)          # :105-107: the division sits inside the loop
        return loss
    if "decoder" in loss_mode:
        with torch.no_grad():
            hr_f = unet_forward(hr, sd, loss_mode, bn)
        return pixel_loss(loss_mode)(sr_f, hr_f)
    if loss_mode == "label-hr":
        with torch.no_grad():
            hr_l = unet_forward(hr, sd, loss_mode, bn)
        return dice_loss_multiclass(sr_f, torch.argmax(hr_l, dim=1), dice_classes)
    if loss_mode == "label-gt":
        if gt_label.dim() == 4:
            gt_label = gt_label[:, 0]
        return dice_loss_multiclass(sr_f, gt_label.to(torch.long), dice_classes)
    raise ValueError("Invalid UNet Seg Loss Mode: {}".format(loss_mode))
