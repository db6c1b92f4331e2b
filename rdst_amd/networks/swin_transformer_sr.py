"""Swin primitives of the RDST hot path on HIP kernels.

Counterpart of the reference's ``networks/swin_transformer_sr.py`` for the symbols on the hot path:
``Mlp`` :13-29, ``window_partition`` :32-43, ``window_reverse`` :46-59, ``WindowAttention`` :62-141,
``SwinTransformerBlock`` :160-274, ``BasicLayer`` :343-398, ``PatchEmbed`` :487-519,
``PatchUnEmbed`` :529-555.  Constructor signatures, attribute / parameter / buffer names and shapes
follow the reference so state dicts load strictly in both directions; the forward bodies are new:
each block is five fused HIP ops (LN+qkv, window attention, proj+residual, LN+fc1, GELU+fc2+residual)
on token-major rows, with the cyclic shift, window partition/reverse and the shifted-window mask
folded into the attention kernel's index math.  No ``timm`` dependency.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _drop_p(p, what):
    """A dropout probability as the reference's nn.Dropout takes it ([0, 1]; 1 would divide by zero in the kernels)."""
    p = float(p or 0.)
    if not (0. <= p < 1.):
        raise ValueError(f"rdst_amd: {what}={p} must be in [0, 1)")
    return p


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath, which networks/swin_transformer_sr.py:8 imports): in
    training a residual branch is dropped for a whole sample with probability ``drop_prob`` and the kept ones are
    scaled by 1 / (1 - drop_prob); identity in eval().  timm itself is not in this image and the random stream is
    device-specific anyway: parity with the reference is unpinned for the DRAWS, the arithmetic given a mask is tested."""

    def __init__(self, drop_prob=0., scale_by_keep=True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def mask(self, x):
        """Per-sample keep mask, drawn AND scaled in fp32 whatever the activation dtype: in bf16 1 / keep would be
        rounded (1 / 0.9 -> 1.109375, -0.16 %) and every kept branch would carry that bias (E[mask] != 1)."""
        keep = 1.0 - self.drop_prob
        m = torch.empty((x.shape[0],) + (1,) * (x.ndim - 1), dtype=torch.float32, device=x.device).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            m.div_(keep)
        return m

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        return (x.float() * self.mask(x)).to(x.dtype)    # fp32 product, one rounding back to the activation dtype

    def extra_repr(self):
        return f"drop_prob={round(self.drop_prob, 3):0.3f}"


def _ln_params(norm):
    """(weight, bias) of an nn.LayerNorm, or (None, None) for nn.Identity (rdst_layer_norm False)."""
    if isinstance(norm, nn.LayerNorm):
        if abs(norm.eps - 1e-5) > 0 or not norm.elementwise_affine:
            raise NotImplementedError("rdst_amd: LayerNorm must be affine with eps 1e-5")
        return norm.weight, norm.bias
    if isinstance(norm, nn.Identity) or norm is None:
        return None, None
    raise NotImplementedError(f"rdst_amd: norm layer {type(norm).__name__} (nn.LayerNorm or nn.Identity)")


def _norm_only(x, norm, out_scale=1.0, residual=None, out_slot=None):
    w, b = _ln_params(norm)
    if w is None:
        y = x if out_scale == 1.0 else x * out_scale
        y = y if residual is None else y + residual
        if out_slot is not None:
            raise NotImplementedError("rdst_amd: dense slot without a LayerNorm")
        return y
    return ops.ln_linear(x, w, b, None, None, out_scale=out_scale, residual=residual, out_slot=out_slot)


class Mlp(nn.Module):
    """fc2(GELU(fc1(x))): fc1 writes the pre-activation, fc2 reads it through GELU (K3)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU:
            raise NotImplementedError("rdst_amd Mlp: only nn.GELU (exact erf form)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(_drop_p(drop, "drop"))

    def forward(self, x, norm=None, residual=None):
        w, b = _ln_params(norm)
        h = ops.ln_linear(x, w, b, self.fc1.weight, self.fc1.bias)
        if self.training and self.drop.p > 0.:
            # swin_transformer_sr.py:24-28 with drop > 0: the activation is materialised so that the mask can sit between it
            # and fc2 (and behind fc2); elementwise torch ops on the device tensors, the Linears stay on the HIP kernels
            h = self.drop(F.gelu(h))
            y = self.drop(ops.ln_linear(h, None, None, self.fc2.weight, self.fc2.bias))
            return y if residual is None else y + residual
        return ops.ln_linear(h, None, None, self.fc2.weight, self.fc2.bias, in_act=ops.ACT_GELU, residual=residual)


def window_partition(x, window_size):
    """(B,H,W,C) -> (num_windows*B, ws, ws, C).  Layout utility kept for API parity; the hot path
    never materialises windows (the attention kernel indexes them in place)."""
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    """(num_windows*B, ws, ws, C) -> (B,H,W,C)."""
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def _relative_position_index(wh, ww):
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    y, x = ys.reshape(-1), xs.reshape(-1)
    return (y[:, None] - y[None, :] + wh - 1) * (2 * ww - 1) + (x[:, None] - x[None, :] + ww - 1)


class WindowAttention(nn.Module):
    """Window multi-head self attention with relative position bias (W-MSA / SW-MSA)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.window_size = window_size  # (Wh, Ww)
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        if window_size[0] != window_size[1]:
            raise NotImplementedError("rdst_amd WindowAttention: square windows only")
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(window_size[0], window_size[1]))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(_drop_p(attn_drop, "attn_drop"))
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(_drop_p(proj_drop, "proj_drop"))
        trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x, mask=None):
        """x: (num_windows*B, N, C) pre-partitioned windows; mask: (num_windows, N, N) or None.
        (Standalone API of the reference; SwinTransformerBlock below uses the un-partitioned fused path.)"""
        B_, N, C = x.shape
        ws = self.window_size[0]
        qkv = ops.ln_linear(x, None, None, self.qkv.weight, self.qkv.bias)
        a = ops.window_attention(qkv.view(B_, ws, ws, 3 * C), self.relative_position_bias_table, ws, ws,
                                 self.num_heads, ws, 0, self.scale, mask=mask, attn_drop=self.active_attn_drop())
        return self.proj_drop(ops.ln_linear(a.view(B_, N, C), None, None, self.proj.weight, self.proj.bias))

    def active_attn_drop(self):
        """attn_drop as it acts now: nn.Dropout is the identity in eval()."""
        return self.attn_drop.p if self.training else 0.

    def extra_repr(self) -> str:
        return f'dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}'


class SwinTransformerBlock(nn.Module):
    """x + Attn(LN(x)) then x + MLP(LN(x)) as five fused HIP ops."""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            # window not smaller than the (constructor) resolution: one window, no shift
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"

        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(
            dim, window_size=to_2tuple(self.window_size), num_heads=num_heads,
            qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path and drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        # kept for strict state-dict parity; the kernel derives the mask from (H, W, ws, shift)
        self.register_buffer("attn_mask", self.calculate_mask(self.input_resolution) if self.shift_size > 0 else None)

    def calculate_mask(self, x_size):
        H, W = x_size
        ws, s = self.window_size, self.shift_size
        rid = torch.zeros(H, dtype=torch.float32)
        rid[H - ws:H - s] = 1
        rid[H - s:] = 2
        cid = torch.zeros(W, dtype=torch.float32)
        cid[W - ws:W - s] = 1
        cid[W - s:] = 2
        img = (rid[:, None] * 3 + cid[None, :]).view(1, H, W, 1)
        mw = window_partition(img, ws).view(-1, ws * ws)
        diff = mw.unsqueeze(1) - mw.unsqueeze(2)
        return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))

    def fuses_input_gradient(self):
        """True when forward() runs as the single autograd node that can add a dense join's gradient slice inside its
        backward (ops.GradSink): no stochastic depth in training, qkv bias present."""
        if self._stochastic():
            return False
        return self.attn.qkv.bias is not None

    def _stochastic(self):
        """True when a random mask (stochastic depth or any dropout) acts in this call: training only."""
        if not self.training:
            return False
        dp = isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob > 0.
        return dp or self.attn.attn_drop.p > 0. or self.attn.proj_drop.p > 0. or self.mlp.drop.p > 0.

    def forward(self, x, x_size, sink=None):
        H, W = x_size
        n1w, n1b = _ln_params(self.norm1)
        n2w, n2b = _ln_params(self.norm2)
        at, mlp = self.attn, self.mlp
        if self._stochastic():
            # stochastic depth (swin_transformer_sr.py:268, :272) and the dropouts (:136 on the attention weights, :140 behind
            # proj, :26 / :28 inside the Mlp): the masks sit between the pieces of the block, so it runs as its op-level
            # chain and the two residual adds are plain tensor ops
            qkv = ops.ln_linear(x, n1w, n1b, at.qkv.weight, at.qkv.bias)
            a = ops.window_attention(qkv, at.relative_position_bias_table, H, W, self.num_heads, self.window_size,
                                     self.shift_size, at.scale, attn_drop=at.active_attn_drop())
            x = x + self.drop_path(at.proj_drop(ops.ln_linear(a, None, None, at.proj.weight, at.proj.bias)))
            return x + self.drop_path(self.mlp(x, norm=self.norm2))
        if at.qkv.bias is None:  # qkv_bias=False: fall back to the op-level chain
            qkv = ops.ln_linear(x, n1w, n1b, at.qkv.weight, None)
            a = ops.window_attention(qkv, at.relative_position_bias_table, H, W, self.num_heads, self.window_size,
                                     self.shift_size, at.scale)
            x = ops.ln_linear(a, None, None, at.proj.weight, at.proj.bias, residual=x)
            return self.mlp(x, norm=self.norm2, residual=x)
        return ops.swin_block(x, n1w, n1b, at.qkv.weight, at.qkv.bias, at.relative_position_bias_table, at.proj.weight,
                              at.proj.bias, n2w, n2b, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, H, W,
                              self.num_heads, self.window_size, self.shift_size, at.scale, sink=sink)

    def extra_repr(self) -> str:
        return f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, " \
               f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}"


class BasicLayer(nn.Module):
    """`depth` Swin blocks, shift 0 for even / window_size//2 for odd positions."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        if downsample is not None:
            raise NotImplementedError("rdst_amd BasicLayer: downsample is not used on the RDST path")
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads,
                                 window_size=window_size, shift_size=0 if (i % 2 == 0) else window_size // 2,
                                 mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop,
                                 attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = None

    def forward(self, x, x_size, sink=None):
        """sink (ops.GradSink, optional): handed to the FIRST block, the consumer of x (see DenseSTLayer.forward_dense)."""
        for i, blk in enumerate(self.blocks):
            x = blk(x, x_size, sink=sink) if (i == 0 and sink is not None) else blk(x, x_size)
        return x

    def extra_repr(self) -> str:
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"


class _PatchBase(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans = in_chans
        self.embed_dim = embed_dim


class PatchEmbed(_PatchBase):
    """NCHW image -> (B, H*W, C) tokens (+ optional LayerNorm); patch_size is only recorded."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__(img_size, patch_size, in_chans, embed_dim, norm_layer)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        B, C, H, W = x.shape
        t = ops.nchw_to_rows(x, torch.float32).view(B, H * W, C)
        return t if self.norm is None else _norm_only(t, self.norm)


class PatchUnEmbed(_PatchBase):
    """(B, H*W, C) tokens -> NCHW image."""

    def forward(self, x, x_size):
        B, HW, C = x.shape
        return ops.rows_to_nchw(x.view(B, x_size[0], x_size[1], C))


# ----------------------------------------------------------------------------------------------------
# SwinIR baseline on the same primitives ("next" row N4 of the hot-path scope):
# RSTB :412-484, Upsample :562-580, UpsampleOneStep :583-602, SwinIR :605-826, swinir_make_model :829-868.
# ----------------------------------------------------------------------------------------------------
import math  # noqa: E402

from .common import Conv2d  # noqa: E402


def _leaky_code(m):
    if isinstance(m, nn.LeakyReLU):
        if abs(m.negative_slope - 0.2) < 1e-12:
            return ops.ACT_LEAKY02
        if abs(m.negative_slope - 0.01) < 1e-12:
            return ops.ACT_LEAKY001
    raise NotImplementedError(f"rdst_amd: activation {m} between convs (LeakyReLU 0.2 / 0.01)")


def _conv_chain_rows(mod, x, residual=None, out_scale=1.0):
    """A Conv2d, or a Sequential of Conv2d / LeakyReLU as the reference builds for '3conv', applied on rows;
    each LeakyReLU rides on the NEXT conv's input load; a trailing activation is returned as a pending code."""
    if isinstance(mod, Conv2d):
        return mod.forward_rows(x, residual=residual, out_scale=out_scale), ops.ACT_NONE
    mods = list(mod)
    pending = ops.ACT_NONE
    n_conv = sum(isinstance(m, Conv2d) for m in mods)
    seen = 0
    for m in mods:
        if isinstance(m, Conv2d):
            seen += 1
            last = seen == n_conv
            x = m.forward_rows(x, in_act=pending, residual=residual if last else None,
                               out_scale=out_scale if last else 1.0)
            pending = ops.ACT_NONE
        else:
            pending = _leaky_code(m)
    return x, pending


def _res_conv(resi_connection, dim):
    if resi_connection == '1conv':
        return Conv2d(dim, dim, 3, 1, 1)
    if resi_connection == '3conv':
        return nn.Sequential(Conv2d(dim, dim // 4, 3, 1, 1), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                             Conv2d(dim // 4, dim // 4, 1, 1, 0), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                             Conv2d(dim // 4, dim, 3, 1, 1))
    return None


class RSTB(nn.Module):
    """Residual Swin Transformer Block: BasicLayer + conv + residual."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False,
                 img_size=224, patch_size=4, resi_connection='1conv'):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.residual_group = BasicLayer(dim=dim, input_resolution=input_resolution, depth=depth, num_heads=num_heads,
                                         window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                         qk_scale=qk_scale, drop=drop, attn_drop=attn_drop, drop_path=drop_path,
                                         norm_layer=norm_layer, downsample=downsample, use_checkpoint=use_checkpoint)
        conv = _res_conv(resi_connection, dim)
        if conv is not None:
            self.conv = conv
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=dim, norm_layer=None)
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=dim,
                                          norm_layer=None)

    def forward(self, x, x_size):
        B, L, C = x.shape
        H, W = x_size
        y = self.residual_group(x, x_size)
        out, _ = _conv_chain_rows(self.conv, y.view(B, H, W, C), residual=x.view(B, H, W, C))
        return out.view(B, L, C)


class Upsample(nn.Sequential):
    """conv(n, 4n) + PixelShuffle(2), log2(scale) times (or conv(n, 9n) + PixelShuffle(3))."""

    def __init__(self, scale, num_feat):
        m = []
        if (scale & (scale - 1)) == 0:
            for _ in range(int(math.log(scale, 2))):
                m.append(Conv2d(num_feat, 4 * num_feat, 3, 1, 1))
                m.append(nn.PixelShuffle(2))
        elif scale == 3:
            m.append(Conv2d(num_feat, 9 * num_feat, 3, 1, 1))
            m.append(nn.PixelShuffle(3))
        else:
            raise ValueError(f'scale {scale} is not supported. ' 'Supported scales: 2^n and 3.')
        super().__init__(*m)

    def forward_rows(self, x, in_act=ops.ACT_NONE):
        mods = list(self)
        for i in range(0, len(mods), 2):
            x = mods[i].forward_rows(x, in_act=in_act, shuffle=mods[i + 1].upscale_factor)
            in_act = ops.ACT_NONE
        return x


class UpsampleOneStep(nn.Sequential):
    """One conv + one PixelShuffle (lightweight SR)."""

    def __init__(self, scale, num_feat, num_out_ch, input_resolution=None):
        self.num_feat = num_feat
        self.input_resolution = input_resolution
        super().__init__(Conv2d(num_feat, (scale ** 2) * num_out_ch, 3, 1, 1), nn.PixelShuffle(scale))

    def forward_rows(self, x, in_act=ops.ACT_NONE):
        return self[0].forward_rows(x, in_act=in_act, shuffle=self[1].upscale_factor)


class SwinIR(nn.Module):
    """SwinIR (the paper's comparison baseline) on the HIP primitives.  Stochastic depth (drop_path_rate, linearly
    increasing over the blocks as in swin_transformer_sr.py:695) is active in train() and the identity in eval()."""

    def __init__(self, img_size=64, patch_size=1, in_chans=3,
                 embed_dim=96, depths=[6, 6, 6, 6], num_heads=[6, 6, 6, 6],
                 window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, upscale=2, img_range=1., upsampler='', resi_connection='1conv',
                 **kwargs):
        super().__init__()
        num_in_ch = in_chans
        num_out_ch = in_chans
        num_feat = 64
        self.img_range = img_range
        if in_chans == 3:
            self.mean = torch.Tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)
        else:
            self.mean = torch.zeros(1, 1, 1, 1)
        self.upscale = upscale
        self.upsampler = upsampler
        self.drop_path_rate = drop_path_rate
        self.conv_first = Conv2d(num_in_ch, embed_dim, 3, 1, 1)
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.ape = ape
        self.patch_norm = patch_norm
        self.num_features = embed_dim
        self.mlp_ratio = mlp_ratio
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim, embed_dim=embed_dim,
                                      norm_layer=norm_layer if self.patch_norm else None)
        num_patches = self.patch_embed.num_patches
        patches_resolution = self.patch_embed.patches_resolution
        self.patches_resolution = patches_resolution
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim,
                                          embed_dim=embed_dim, norm_layer=norm_layer if self.patch_norm else None)
        if self.ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
            trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=_drop_p(drop_rate, "drop_rate"))
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]   # stochastic depth decay rule
        self.layers = nn.ModuleList()
        for i_layer in range(self.num_layers):
            self.layers.append(RSTB(dim=embed_dim, input_resolution=(patches_resolution[0], patches_resolution[1]),
                                    depth=depths[i_layer], num_heads=num_heads[i_layer], window_size=window_size,
                                    mlp_ratio=self.mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                                    attn_drop=attn_drop_rate,
                                    drop_path=dpr[sum(depths[:i_layer]):sum(depths[:i_layer + 1])],
                                    norm_layer=norm_layer, downsample=None,
                                    use_checkpoint=use_checkpoint, img_size=img_size, patch_size=patch_size,
                                    resi_connection=resi_connection))
        self.norm = norm_layer(self.num_features)
        cab = _res_conv(resi_connection, embed_dim)
        if cab is not None:
            self.conv_after_body = cab
        if self.upsampler == 'pixelshuffle':
            self.conv_before_upsample = nn.Sequential(Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.upsample = Upsample(upscale, num_feat)
            self.conv_last = Conv2d(num_feat, num_out_ch, 3, 1, 1)
        elif self.upsampler == 'pixelshuffledirect':
            self.upsample = UpsampleOneStep(upscale, embed_dim, num_out_ch, (patches_resolution[0], patches_resolution[1]))
        elif self.upsampler == 'nearest+conv':   # real-world SR (swin_transformer_sr.py:735-745): x4 only
            assert self.upscale == 4, 'only support x4 now.'
            self.conv_before_upsample = nn.Sequential(Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.conv_up1 = Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_up2 = Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_hr = Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_last = Conv2d(num_feat, num_out_ch, 3, 1, 1)
            self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        else:
            self.conv_last = Conv2d(embed_dim, num_out_ch, 3, 1, 1)
        self.compute_dtype, self.compute_code = torch.float32, ops.F32
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def set_compute_dtype(self, dtype):
        """torch.float32 / 'fp32' exact, 'fp32x3' split-bf16 GEMMs on fp32 tensors, torch.bfloat16 / 'bf16': this module's own mode."""
        self.compute_dtype, self.compute_code = ops.resolve_compute_dtype(dtype)
        return self

    def _features_rows(self, feat):
        B, H, W, E = feat.shape
        t = feat.view(B, H * W, E)
        if self.patch_embed.norm is not None:
            t = _norm_only(t, self.patch_embed.norm)
        if self.ape:
            t = t + self.absolute_pos_embed.to(t.dtype)
        t = self.pos_drop(t)   # swin_transformer_sr.py:774 (identity unless drop_rate > 0 in training)
        for layer in self.layers:
            t = layer(t, (H, W))
        return _norm_only(t, self.norm).view(B, H, W, E)

    def forward(self, x):
        with ops.compute_scope(self.compute_code):
            return self._forward(x)

    def _forward(self, x):
        self.mean = self.mean.type_as(x)
        xin = (x - self.mean) * self.img_range
        rows = ops.nchw_to_rows(xin, self.compute_dtype)
        feat = self.conv_first.forward_rows(rows)
        res, pend = _conv_chain_rows(self.conv_after_body, self._features_rows(feat), residual=feat)
        if self.upsampler == 'pixelshuffle':
            y, pend = _conv_chain_rows(self.conv_before_upsample, res)
            y = self.upsample.forward_rows(y, in_act=pend)
            y = self.conv_last.forward_rows(y)
        elif self.upsampler == 'pixelshuffledirect':
            y = self.upsample.forward_rows(res)
        elif self.upsampler == 'nearest+conv':
            # swin_transformer_sr.py:799-806.  A pointwise activation commutes with nearest-neighbour replication, so every
            # LeakyReLU rides on the input load of the conv behind the upsampling (no activation tensor is written).
            y, pend = _conv_chain_rows(self.conv_before_upsample, res)
            lr = _leaky_code(self.lrelu)
            y = self.conv_up1.forward_rows(ops.upsample_nearest2(y), in_act=pend)
            y = self.conv_up2.forward_rows(ops.upsample_nearest2(y), in_act=lr)
            y = self.conv_hr.forward_rows(y, in_act=lr)
            y = self.conv_last.forward_rows(y, in_act=lr)
        else:
            y = self.conv_last.forward_rows(res, residual=rows)
        return ops.rows_to_nchw(y) / self.img_range + self.mean


def swinir_make_model(paras):
    upscale = paras.sr_scale
    window_size = paras.sir_window_size
    img_size = int(paras.patch_size // upscale // window_size + 1) * window_size
    return SwinIR(
        img_size=img_size, patch_size=paras.sir_token_size, in_chans=paras.input_channel,
        embed_dim=paras.sir_embed_dim, depths=paras.sir_swintr_layers, num_heads=paras.sir_num_heads,
        window_size=window_size, mlp_ratio=paras.sir_hidden_ratio, qkv_bias=paras.sir_qkv_bias,
        qk_scale=paras.sir_qk_scale, drop_rate=paras.sir_drop_rate, attn_drop_rate=paras.sir_attn_drop_rate,
        drop_path_rate=paras.sir_drop_path_rate, norm_layer=nn.LayerNorm if paras.sir_layer_norm else nn.Identity,
        ape=paras.sir_ape, patch_norm=paras.sir_patch_norm, use_checkpoint=paras.sir_use_checkpoint,
        upscale=int(upscale), img_range=paras.sir_img_range, upsampler=paras.sir_upsampler,
        resi_connection=paras.sir_res_connection)
