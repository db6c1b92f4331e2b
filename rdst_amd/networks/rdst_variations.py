"""RDST network family on HIP kernels: DenseSTLayer, RDSTB, RDSTSR, make_RDSTSR.

Counterpart of the reference's ``networks/rdst_variations.py`` (identical, for these classes, to
``networks/swinIR_variations.py`` which is what its trainer/tester import): ``DenseSTLayer``
:246-341, ``RDSTB`` :354-445, ``RDSTSR`` :1115-1366, ``make_RDSTSR`` :1369-1457.  Constructor and
``forward`` signatures, accepted-but-inert arguments, validation errors and parameter names follow
the reference; ``state_dict()`` keys/shapes are identical (tests/golden/state_dict_*.json).

The forward is new.  Activations stay token-major ("rows", = NHWC) from the head conv to the last
conv, so every PatchEmbed/PatchUnEmbed transpose of the reference disappears; each op is a fused
HIP kernel from librdst_hip.so (see include/rdst_hip.h).  ``compute_dtype`` selects fp32 (parity
mode, default) or bf16 activations (throughput mode; parameters stay fp32).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .common import Conv2d, MeanShift, UpSampler, default_conv
from .swin_transformer_sr import (BasicLayer, Mlp, PatchEmbed, PatchUnEmbed, SwinTransformerBlock,  # noqa: F401
                                  WindowAttention, _drop_p, _ln_params, _norm_only, trunc_normal_, window_partition,
                                  window_reverse)


def _dim_modifier(input_dim, growth_rate, norm_layer, pre_norm):
    """Sequential(LN(in), Linear(in, growth)) if pre_norm else Sequential(Linear, LN(growth))."""
    if pre_norm:
        return nn.Sequential(norm_layer(input_dim), nn.Linear(input_dim, growth_rate))
    return nn.Sequential(nn.Linear(input_dim, growth_rate), norm_layer(growth_rate))


def _apply_dim_modifier(seq, x, out_scale=1.0, out_slot=None):
    """out_slot = (ops.DenseBuffer, first channel): the last op writes straight into the RDSTB's dense buffer."""
    if isinstance(seq, nn.Identity):
        if out_slot is not None:
            raise NotImplementedError("rdst_amd: dense slot behind an Identity tail")
        return x if out_scale == 1.0 else x * out_scale
    a, b = seq[0], seq[1]
    if isinstance(b, nn.Linear):  # (norm, linear): one fused op
        w, bb = _ln_params(a)
        return ops.ln_linear(x, w, bb, b.weight, b.bias, out_scale=out_scale, out_slot=out_slot)
    y = ops.ln_linear(x, None, None, a.weight, a.bias)  # (linear, norm)
    return _norm_only(y, b, out_scale=out_scale, out_slot=out_slot)


def _slot_ok(seq):
    """Can this tail write into a dense slot?  (Needs a fused op at its end: a Linear, or a LayerNorm that is not Identity.)"""
    if isinstance(seq, nn.Identity):
        return False
    b = seq[1]
    return isinstance(b, nn.Linear) or isinstance(b, nn.LayerNorm)


class DenseSTLayer(nn.Module):
    """x -> cat(x, dense_scale * tail(BasicLayer(head(x)))) along channels."""

    def __init__(self, input_dim, input_resolution, depth=2, num_heads=6, window_size=2,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=30., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False,
                 growth_rate=60, dense_scale=1., dim_modify_mode='tail',
                 pre_norm=False):
        super().__init__()
        assert growth_rate % num_heads == 0, 'growth_rate % num_heads should be 0'
        assert input_dim % num_heads == 0, 'token dim % num_heads should be 0'
        if dim_modify_mode == 'head':
            self.head = (_dim_modifier(input_dim, growth_rate, norm_layer, pre_norm)
                         if input_dim != growth_rate else nn.Identity())
            hidden_dim = growth_rate
            self.tail = nn.Identity()
        elif dim_modify_mode == 'tail':
            self.head = nn.Identity()
            hidden_dim = input_dim
            self.tail = (_dim_modifier(hidden_dim, growth_rate, norm_layer, pre_norm)
                         if hidden_dim != growth_rate else nn.Identity())
        else:
            raise ValueError(f"dim_modify_mode {dim_modify_mode!r} (head or tail)")
        # the reference's own default drop_path=30. can only ever be used as "no drop path"
        # (RDSTB always passes its own value, 0. on every shipped path)
        self.body = BasicLayer(hidden_dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias,
                               qk_scale, drop, attn_drop, drop_path, norm_layer, downsample, use_checkpoint)
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.input_resolution = input_resolution
        self.dense_scale = dense_scale
        self.growth_rate = hidden_dim
        self.growth_rate_out = growth_rate   # channels forward() appends (not a reference attribute)
        self.depth = depth
        self.pre_norm = pre_norm

    def new_features(self, x, x_size, out_slot=None, sink=None):
        """dense_scale * tail(body(head(x))): the channels this layer appends."""
        y = _apply_dim_modifier(self.head, x)
        y = self.body(y, x_size, sink=sink) if sink is not None else self.body(y, x_size)
        return _apply_dim_modifier(self.tail, y, out_scale=self.dense_scale, out_slot=out_slot)

    def _grad_sink(self, x):
        """An ops.GradSink when the body's first Swin block is the only other consumer of x and takes the join's
        gradient slice inside its backward (head = Identity, the block runs as one autograd node); else None."""
        blocks = self.body.blocks
        if isinstance(self.head, nn.Identity) and len(blocks) > 0 and blocks[0].fuses_input_gradient() \
                and x.requires_grad and torch.is_grad_enabled():
            return ops.GradSink()
        return None

    def forward(self, x, x_size):
        return torch.cat((x, self.new_features(x, x_size)), 2)

    def forward_dense(self, x, x_size, buf):
        """The same with x already lying in the first channels of the RDSTB's dense buffer: the new channels are
        written next to it in place and the wider view is returned (no copy of x)."""
        sink = self._grad_sink(x)
        if not _slot_ok(self.tail):
            new = self.new_features(x, x_size, sink=sink)
            buf.fill(x.shape[-1], new)   # values only: `new` itself stays in the graph
            return ops.dense_join(x, new, buf, sink)
        new = self.new_features(x, x_size, out_slot=(buf, x.shape[-1]), sink=sink)
        return ops.dense_join(x, new, buf, sink)


def _res_connection(resi_connection, cin, cout):
    if resi_connection == '1conv':
        return Conv2d(cin, cout, 3, 1, 1)
    if resi_connection == '3conv':
        return nn.Sequential(Conv2d(cin, cin // 4, 3, 1, 1),
                             nn.LeakyReLU(negative_slope=0.2, inplace=True),
                             Conv2d(cin // 4, cin // 4, 1, 1, 0),
                             nn.LeakyReLU(negative_slope=0.2, inplace=True),
                             Conv2d(cin // 4, cout, 3, 1, 1))
    return None


def _apply_res_connection(conv, x, residual=None, out_scale=1.0, out_slot=None):
    """conv(x) * out_scale + residual on rows; the LeakyReLUs of '3conv' ride on the next conv's load.
    out_slot = (DenseBuffer, channel): the last conv writes there (the next RDSTB's dense buffer)."""
    if isinstance(conv, Conv2d):
        return conv.forward_rows(x, residual=residual, out_scale=out_scale, out_slot=out_slot)
    y = conv[0].forward_rows(x)
    y = conv[2].forward_rows(y, in_act=ops.ACT_LEAKY02)
    return conv[4].forward_rows(y, in_act=ops.ACT_LEAKY02, residual=residual, out_scale=out_scale, out_slot=out_slot)


class RDSTB(nn.Module):
    """Residual Dense Swin Transformer Block: num_blocks DenseSTLayers, 3x3 fusion conv, residual."""

    def __init__(self, input_dim, input_resolution, layer_depth, num_heads=6, window_size=2,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False,
                 img_size=224, patch_size=4, resi_connection='1conv',
                 growth_rate=0, dense_scale=1., dim_modify_mode='tail',
                 num_blocks=3, residual_scale=1.,
                 pre_norm=False):
        super().__init__()
        self.input_dim = input_dim
        self.input_resolution = input_resolution
        self.residual_scale = residual_scale
        idim = input_dim
        self.body = nn.ModuleList([])
        for _ in range(int(num_blocks)):
            self.body.append(DenseSTLayer(
                input_dim=idim, input_resolution=input_resolution, depth=layer_depth, num_heads=num_heads,
                window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop,
                attn_drop=attn_drop, drop_path=drop_path, norm_layer=norm_layer, downsample=downsample,
                use_checkpoint=use_checkpoint, growth_rate=growth_rate, dense_scale=dense_scale,
                dim_modify_mode=dim_modify_mode, pre_norm=pre_norm))
            idim += growth_rate
        conv = _res_connection(resi_connection, idim, input_dim)
        if conv is not None:
            self.conv = conv
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=input_dim,
                                      norm_layer=None)
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=idim,
                                          norm_layer=None)

    def dense_width(self):
        return self.input_dim + sum(m.growth_rate_out for m in self.body)

    def make_buffer(self, B, L, dtype, device):
        """The dense buffer forward() works in; a caller that creates it ahead can have the PRODUCER of x write x into
        its first channels (out_slot = (buf, 0)), which spares the one copy an RDSTB makes."""
        return ops.DenseBuffer((B, L), self.dense_width(), dtype, device)

    def forward(self, x, x_size, buf=None, out_slot=None):
        """buf: this block's dense buffer (make_buffer), x possibly already in its first channels; out_slot: where the
        fusion conv writes the block's output (the next block's buffer)."""
        B, L, C = x.shape
        H, W = x_size
        if x.is_cuda and len(self.body) > 0:
            # dense buffer: every DenseSTLayer appends its channels in place (no torch.cat of the growing prefix)
            if buf is None:
                buf = self.make_buffer(B, L, x.dtype, x.device)
            y = ops.into_dense(x, buf)
            for m in self.body:
                y = m.forward_dense(y, x_size, buf)
        else:
            y = x
            for m in self.body:
                y = m(y, x_size)
        out = _apply_res_connection(self.conv, y.view(B, H, W, y.shape[-1]), residual=x.view(B, H, W, C),
                                    out_scale=self.residual_scale, out_slot=out_slot)
        return out.view(B, L, C)


class RDSTSR(nn.Module):
    """MeanShift + head conv + n x RDSTB + LayerNorm + conv + global residual + pixel-shuffle tail."""

    def __init__(self, img_size=48, patch_size=1, in_chans=1, sr_scale=2, embed_dim=60,
                 dense_layer_depths=[2, 2, 2, 2], num_heads=[6, 6, 6, 6],
                 window_size=[4, 4, 4, 4], rdb_depths=[3, 3, 3, 3],
                 mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, resi_connection='1conv',
                 growth_rate=30, dense_scale=1., dim_modify_mode='tail',
                 rdb_residual_scale=1.,
                 global_res_scale=1.,
                 mean=None, std=None,
                 act_in_conv='leaky_relu', bn_in_conv=None,
                 scale_free=False,
                 scale_embedding=False,
                 pre_norm=False,
                 feature_last_operation=False):
        super().__init__()
        self.input_resolution = img_size
        self.patch_size = patch_size
        self.input_channel = in_chans
        self.num_blocks = len(rdb_depths)
        assert len(rdb_depths) == len(window_size) == len(num_heads) == len(dense_layer_depths)
        self.n_feats = embed_dim
        self.patch_norm = patch_norm
        self.ape = ape
        self.use_checkpoint = use_checkpoint      # accepted, inert (as in the reference)
        self.drop_path_rate = drop_path_rate      # accepted, inert: never forwarded to the blocks
        self.drop_rate = drop_rate
        self.resi_connection = resi_connection
        self.global_res_scale = global_res_scale
        self.window_size = window_size
        self.mlp_ratio = mlp_ratio
        self.qkv_bias = qkv_bias
        self.qk_scale = qk_scale
        self.norm_layer = norm_layer
        self.num_heads = num_heads
        self.dense_layer_depths = dense_layer_depths
        self.dense_scale = dense_scale
        self.growth_rate = growth_rate
        self.dim_modify_mode = dim_modify_mode
        self.rdb_depths = rdb_depths
        self.rdb_residual_scale = rdb_residual_scale

        if act_in_conv == 'relu':
            self.act_in_conv = nn.ReLU(True)
        elif act_in_conv == 'leaky_relu':
            self.act_in_conv = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        elif act_in_conv == 'prelu':
            self.act_in_conv = nn.PReLU()
        else:
            raise ValueError('Invalid activation {}, should be one of [relu, leaky_relu, prelu]'.format(act_in_conv))
        self.bn_in_conv = bn_in_conv
        self.sr_scale = sr_scale
        self.scale_free = scale_free
        self.scale_embedding = scale_embedding
        if scale_embedding:
            self.scale_embedding = nn.Identity()
            self.scale_embed_layer = nn.Linear(1, 1)
        if scale_free:
            raise NotImplementedError("rdst_amd RDSTSR: scale_free=True (MetaUpSampler) is outside the hot path; "
                                      "every shipped config has scale_free = False")

        if mean is None:
            mean = [0. for _ in range(self.input_channel)]
        if std is None:
            std = [1. for _ in range(self.input_channel)]
        if len(mean) != len(std) or len(mean) != self.input_channel:
            raise ValueError('Dimension of mean {} / std {} should fit input channels {}'.format(
                len(mean), len(std), self.input_channel))
        self.mean = mean
        self.std = std
        self.add_mean = MeanShift(mean, std, 'add')
        self.sub_mean = MeanShift(mean, std, 'sub')

        self.head = default_conv(in_chans, embed_dim, 3)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim,
                                      embed_dim=embed_dim, norm_layer=norm_layer if self.patch_norm else None)
        num_patches = self.patch_embed.num_patches
        patches_resolution = self.patch_embed.patches_resolution
        self.patches_resolution = patches_resolution
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim,
                                          embed_dim=embed_dim, norm_layer=norm_layer if self.patch_norm else None)
        if self.ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
            trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=_drop_p(drop_rate, "drop_rate"))

        self.body = nn.ModuleList()
        for i_block in range(self.num_blocks):
            self.body.append(RDSTB(
                input_dim=embed_dim, input_resolution=(patches_resolution[0], patches_resolution[1]),
                layer_depth=self.dense_layer_depths[i_block], num_heads=self.num_heads[i_block],
                window_size=self.window_size[i_block], mlp_ratio=self.mlp_ratio, qkv_bias=self.qkv_bias,
                qk_scale=self.qk_scale, drop=self.drop_rate, attn_drop=attn_drop, norm_layer=norm_layer,
                img_size=self.input_resolution, patch_size=self.patch_size, resi_connection=self.resi_connection,
                growth_rate=self.growth_rate, dense_scale=self.dense_scale, dim_modify_mode=self.dim_modify_mode,
                num_blocks=self.rdb_depths[i_block], residual_scale=self.rdb_residual_scale, pre_norm=pre_norm))
        self.norm = norm_layer(self.n_feats)

        self.feature_last_operation = feature_last_operation
        cab = _res_connection(resi_connection, embed_dim, embed_dim)
        if cab is not None:
            self.conv_after_body = cab

        m_tail = []
        if self.sr_scale > 1:
            m_tail.append(UpSampler(default_conv, self.sr_scale, self.n_feats, act=None, bn=self.bn_in_conv))
        m_tail.append(default_conv(self.n_feats, self.input_channel, 3))
        self.tail = nn.Sequential(*m_tail)

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
        """torch.float32 / 'fp32' = parity mode (default); 'fp32x3' = fp32 tensors, split-bf16 GEMMs; torch.bfloat16 /
        'bf16' = throughput mode (bf16 activations, fp32 accumulation, fp32 parameters and gradients).  The mode is this
        module's own (``compute_code``): it does not touch any other network of the process."""
        self.compute_dtype, self.compute_code = ops.resolve_compute_dtype(dtype)
        return self

    def forward_features_rows(self, feat):
        """feat: rows (B,H,W,E) after the head conv -> rows (B,H,W,E): LN(body(patch_norm(feat))) * global_res_scale
        (+ feat when there is no conv_after_body step to add it)."""
        B, H, W, E = feat.shape
        x_size = (H, W)
        t = feat.view(B, H * W, E)
        # every RDSTB's dense buffer is created ahead, so that the producer of a block's input (the patch norm, the
        # previous block's fusion conv) writes it straight into the buffer's first channels: no copy per block
        blocks = list(self.body)
        chain = t.is_cuda and len(blocks) > 0 and all(isinstance(b, RDSTB) and len(b.body) > 0 and hasattr(b, "conv")
                                                      and b.input_dim == E for b in blocks)
        bufs = [b.make_buffer(B, H * W, t.dtype, t.device) for b in blocks] if chain else [None] * len(blocks)
        pos_drop = self.training and self.pos_drop.p > 0.
        if self.patch_embed.norm is not None:
            first = (bufs[0], 0) if (chain and not self.ape and not pos_drop and isinstance(self.patch_embed.norm, nn.LayerNorm)) else None
            t = _norm_only(t, self.patch_embed.norm, out_slot=first)
        if self.ape:
            t = t + self.absolute_pos_embed.to(t.dtype)
        if pos_drop:
            t = self.pos_drop(t)   # rdst_variations.py:1332
        for i, blk in enumerate(blocks):
            if chain:
                t = blk(t, x_size, buf=bufs[i], out_slot=(bufs[i + 1], 0) if i + 1 < len(blocks) else None)
            else:
                t = blk(t, x_size)
        return t

    def forward_features(self, x):
        """NCHW feature map -> NCHW (API parity with the reference's forward_features)."""
        with ops.compute_scope(self.compute_code):
            rows = ops.nchw_to_rows(x, self.compute_dtype)
            t = self.forward_features_rows(rows)
            B, H, W, E = rows.shape
            return ops.rows_to_nchw(_norm_only(t, self.norm).view(B, H, W, E))

    def forward(self, x, sr_scale=None):
        # the packed weight images of all layers: one batched pack per forward; the arithmetic of THIS network's ops
        with ops.compute_scope(self.compute_code), ops.pack_scope(self):
            return self._forward(x, sr_scale)

    def _forward(self, x, sr_scale=None):
        rows = ops.nchw_to_rows(x, self.compute_dtype)          # (B,H,W,nc)
        rows = self.sub_mean.forward_rows(rows)
        feat = self.head.forward_rows(rows)                      # (B,H,W,E)
        B, H, W, E = feat.shape
        t = self.forward_features_rows(feat)
        if self.feature_last_operation:
            t = _norm_only(t, self.norm, out_scale=self.global_res_scale)
            res = _apply_res_connection(self.conv_after_body, t.view(B, H, W, E), residual=feat)
        else:
            res = _norm_only(t, self.norm, out_scale=self.global_res_scale, residual=feat.view(B, H * W, E))
            res = res.view(B, H, W, E)
        y = res
        for m in self.tail:
            y = m.forward_rows(y)
        y = self.add_mean.forward_rows(y)
        return ops.rows_to_nchw(y)

    def extra_repr(self):
        return ''

    def flops(self):
        return None


class RDSTSR_N(RDSTSR):
    """RDSTSR with an RDN-style global bottleneck: the outputs of all RDSTBs are concatenated and fused by
    Linear -> Linear ('mlp') or conv1x1 -> conv3x3 ('conv').  Reference: rdst_variations.py:824-1112 ("next"
    row N4).  Like the reference's forward (:1090-1106) it uses neither the final ``norm`` nor
    ``conv_after_body`` (both still exist as parameters, for state-dict parity)."""

    def __init__(self, img_size=48, patch_size=1, in_chans=1, sr_scale=2, embed_dim=60,
                 dense_layer_depths=[2, 2, 2, 2], num_heads=[6, 6, 6, 6],
                 window_size=[4, 4, 4, 4], rdb_depths=[3, 3, 3, 3],
                 mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, resi_connection='1conv',
                 growth_rate=30, dense_scale=1., dim_modify_mode='tail',
                 rdb_residual_scale=1., global_res_scale=1., mean=None, std=None,
                 act_in_conv='leaky_relu', bn_in_conv=None, scale_free=False, scale_embedding=False,
                 pre_norm=False, global_bottleneck=True, global_bottleneck_ratio=1., global_bottleneck_mode='mlp'):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, sr_scale=sr_scale,
                         embed_dim=embed_dim, dense_layer_depths=dense_layer_depths, num_heads=num_heads,
                         window_size=window_size, rdb_depths=rdb_depths, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                         qk_scale=qk_scale, drop_rate=drop_rate, attn_drop=attn_drop, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, ape=ape, patch_norm=patch_norm, use_checkpoint=use_checkpoint,
                         resi_connection=resi_connection, growth_rate=growth_rate, dense_scale=dense_scale,
                         dim_modify_mode=dim_modify_mode, rdb_residual_scale=rdb_residual_scale,
                         global_res_scale=global_res_scale, mean=mean, std=std, act_in_conv=act_in_conv,
                         bn_in_conv=bn_in_conv, scale_free=scale_free, scale_embedding=scale_embedding,
                         pre_norm=pre_norm, feature_last_operation=False)
        del self.feature_last_operation
        # the reference registers: ... body, norm, bottleneck, conv_after_body, tail  (rdst_variations.py:984-1046)
        cab = self._modules.pop('conv_after_body', None)
        tail = self._modules.pop('tail')
        self.global_bottleneck_mode = global_bottleneck_mode
        self.do_global_bottleneck = global_bottleneck
        if self.do_global_bottleneck:
            cin = self.n_feats * self.num_blocks
            cf = int(self.n_feats * global_bottleneck_ratio)
            if global_bottleneck_mode == 'mlp':
                self.bottleneck = nn.Sequential(nn.Linear(cin, cf), nn.Linear(cf, cf))
            elif global_bottleneck_mode == 'conv':
                self.bottleneck = nn.Sequential(default_conv(cin, cf, 1), default_conv(cf, cf, 3))
            else:
                raise ValueError(f"global_bottleneck_mode {global_bottleneck_mode!r} (mlp or conv)")
        else:
            cf = self.n_feats
            self.bottleneck = None
        if cab is not None:
            self.conv_after_body = cab
        if cf != self.n_feats:
            m_tail = []
            if self.sr_scale > 1:
                m_tail.append(UpSampler(default_conv, self.sr_scale, cf, act=None, bn=self.bn_in_conv))
            m_tail.append(default_conv(cf, self.input_channel, 3))
            tail = nn.Sequential(*m_tail)
        self.tail = tail
        if self.bottleneck is not None:
            self.bottleneck.apply(self._init_weights)

    def forward(self, x, sr_scale=None):
        with ops.compute_scope(self.compute_code):
            return self._forward_n(x, sr_scale)

    def _forward_n(self, x, sr_scale=None):
        rows = ops.nchw_to_rows(x, self.compute_dtype)
        rows = self.sub_mean.forward_rows(rows)
        feat = self.head.forward_rows(rows)
        B, H, W, E = feat.shape
        t = feat.view(B, H * W, E)
        if self.patch_embed.norm is not None:
            t = _norm_only(t, self.patch_embed.norm)
        if self.ape:
            t = t + self.absolute_pos_embed.to(t.dtype)
        t = self.pos_drop(t)   # rdst_variations.py:1068
        gs = self.global_res_scale
        if self.do_global_bottleneck:
            maps = []
            for blk in self.body:
                t = blk(t, (H, W))
                maps.append(t)
            fm = torch.cat(maps, 2)
            if self.global_bottleneck_mode == 'mlp':
                y = ops.ln_linear(fm, None, None, self.bottleneck[0].weight, self.bottleneck[0].bias)
                res = ops.ln_linear(y, None, None, self.bottleneck[1].weight, self.bottleneck[1].bias, out_scale=gs,
                                    residual=feat.view(B, H * W, E)).view(B, H, W, -1)
            else:
                y = self.bottleneck[0].forward_rows(fm.view(B, H, W, -1))
                res = self.bottleneck[1].forward_rows(y, out_scale=gs, residual=feat)
        else:
            for blk in self.body:
                t = blk(t, (H, W))
            res = (t * gs + feat.view(B, H * W, E)).view(B, H, W, E)
        y = res
        for m in self.tail:
            y = m.forward_rows(y)
        y = self.add_mean.forward_rows(y)
        return ops.rows_to_nchw(y)


def make_RDSTSR(paras, mean=None, std=None):
    """Build the network from a flat parameter namespace (the reference's ParametersLoader)."""
    sr_scale = int(paras.sr_scale)
    norm_layer = nn.LayerNorm if paras.rdst_layer_norm else nn.Identity
    kw = dict(
        img_size=paras.patch_size, patch_size=paras.swin_patch_size, in_chans=paras.input_channel,
        sr_scale=sr_scale, embed_dim=paras.rdst_embed_dim,
        dense_layer_depths=paras.rdst_dense_layer_depths, num_heads=paras.rdst_num_heads,
        window_size=paras.rdst_window_size, rdb_depths=paras.rdst_rdb_depths,
        mlp_ratio=paras.swin_hidden_ratio, qkv_bias=paras.swin_qkv_bias, qk_scale=paras.swin_qk_scale,
        drop_rate=paras.swin_drop_rate, attn_drop=paras.swin_attn_drop_rate,
        drop_path_rate=paras.swin_drop_path_rate,
        norm_layer=norm_layer, ape=paras.rdst_ape, patch_norm=paras.rdst_patch_norm,
        use_checkpoint=paras.rdst_use_checkpoint, resi_connection=paras.rdst_res_connection,
        growth_rate=paras.rdst_growth_rate, dense_scale=paras.rdst_dense_scale,
        dim_modify_mode=paras.rdst_dim_modify_mode,
        rdb_residual_scale=paras.rdst_rdb_residual_scale, global_res_scale=paras.rdst_global_res_scale,
        mean=mean, std=std,
        act_in_conv=paras.rdst_act_in_conv, bn_in_conv=paras.rdst_bn_in_conv,
        scale_free=paras.scale_free, pre_norm=paras.rdst_pre_norm)
    # attributes the reference reads unconditionally (rdst_variations.py:1380-1382)
    global_bottleneck = paras.rdst_global_bottleneck
    _ratio = paras.rdst_global_bottleneck_ratio
    feature_last_operation = paras.rdst_feature_last_operation
    if not global_bottleneck:
        return RDSTSR(feature_last_operation=feature_last_operation, **kw)
    return RDSTSR_N(global_bottleneck=global_bottleneck, global_bottleneck_ratio=_ratio,
                    global_bottleneck_mode=paras.rdst_global_bottleneck_mode, **kw)
