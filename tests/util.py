"""Shared helpers for the test-suite."""
import os

import numpy as np
import torch

from oracle import rdst_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

NET_CASES = {
    "net_tiny_64": (O.CFG_TINY, 1),
    "net_tiny_b4": (O.CFG_TINY, 1),      # BASELINE.json configs[0] at its stated shape 4 x 1 x 64 x 64, all gradients elementwise
    "net_e1_16": (O.CFG_E1, 2),
    "net_e1_eval_40x32": (O.CFG_E1, 2),
    "net_ws16_32": (O.make_cfg(**{**O.CFG_WS16, "img_size": 32, "dense_layer_depths": [2, 2], "num_heads": [6, 6],
                                  "window_size": [16, 16], "rdb_depths": [3, 2]}), 3),
    "net_3conv_x3": (O.make_cfg(img_size=16, in_chans=1, sr_scale=3, embed_dim=48, dense_layer_depths=[2],
                                num_heads=[6], window_size=[8], rdb_depths=[2], mlp_ratio=2.0, growth_rate=24,
                                pre_norm=False, resi_connection="3conv", feature_last_operation=False,
                                dense_scale=0.5, rdb_residual_scale=0.7, global_res_scale=0.9), 4),
}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def rand(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))


def build_net(cfg, mean=None, std=None, **extra):
    """The HIP-backed RDSTSR for an oracle cfg dict (same kwargs the golden generator gave the reference); `extra`: further
    constructor arguments (drop_rate, attn_drop ...)."""
    import torch.nn as nn
    from rdst_amd.networks.rdst_variations import RDSTSR
    return RDSTSR(
        img_size=cfg["img_size"], patch_size=1, in_chans=cfg["in_chans"], sr_scale=cfg["sr_scale"],
        embed_dim=cfg["embed_dim"], dense_layer_depths=cfg["dense_layer_depths"], num_heads=cfg["num_heads"],
        window_size=cfg["window_size"], rdb_depths=cfg["rdb_depths"], mlp_ratio=cfg["mlp_ratio"],
        qk_scale=cfg["qk_scale"], norm_layer=nn.LayerNorm if cfg["layer_norm"] else nn.Identity,
        patch_norm=cfg["patch_norm"], resi_connection=cfg["resi_connection"], growth_rate=cfg["growth_rate"],
        dense_scale=cfg["dense_scale"], rdb_residual_scale=cfg["rdb_residual_scale"],
        global_res_scale=cfg["global_res_scale"], mean=mean, std=std, pre_norm=cfg["pre_norm"],
        feature_last_operation=cfg["feature_last_operation"], **extra)


def seeded_fill(state_dict, seed):
    """Deterministic weights for ANY module, keyed by state-dict name (numpy PCG64): the golden generator
    fills the reference with it, the tests fill the HIP module with it; integer buffers, masks and MeanShift
    tensors keep their constructed values."""
    import zlib
    out = {}
    for k, v in state_dict.items():
        if (not v.dtype.is_floating_point) or k.endswith("attn_mask") or k.startswith(("sub_mean.", "add_mean.")):
            out[k] = v.clone()
            continue
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(k.encode())]))
        n = torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32))
        if k.endswith("relative_position_bias_table"):
            a = 0.5 * n
        elif v.dim() == 1 and k.endswith("weight"):
            a = 1.0 + 0.1 * n            # LayerNorm weight
        elif v.dim() == 1:
            a = 0.05 * n                 # biases
        elif v.dim() == 4:
            a = n / float(np.sqrt(v.shape[1] * v.shape[2] * v.shape[3]))
        elif v.dim() == 2:
            a = 0.7 * n / float(np.sqrt(v.shape[1]))
        else:
            a = 0.02 * n
        out[k] = a
    return out


MODEL_CASES = {
    # name: (kind, ctor kwargs, input shape, seed, train?)
    "swinir_ps_x4": ("swinir", dict(img_size=16, in_chans=1, embed_dim=60, depths=[2, 2], num_heads=[6, 6], window_size=8,
                                    mlp_ratio=2., upscale=4, img_range=1., upsampler="pixelshuffle", drop_path_rate=0.),
                     (2, 1, 16, 16), 21, True),
    "swinir_psd_x2_rgb": ("swinir", dict(img_size=16, in_chans=3, embed_dim=48, depths=[2], num_heads=[6], window_size=8,
                                         mlp_ratio=2., upscale=2, img_range=255., upsampler="pixelshuffledirect",
                                         drop_path_rate=0.1, resi_connection="3conv"),
                          (1, 3, 16, 24), 22, False),
    "swinir_denoise": ("swinir", dict(img_size=16, in_chans=1, embed_dim=48, depths=[2], num_heads=[6], window_size=8,
                                      mlp_ratio=2., upscale=1, img_range=1., upsampler="", drop_path_rate=0.),
                       (1, 1, 16, 16), 23, True),
    "swinir_nearest_x4": ("swinir", dict(img_size=16, in_chans=3, embed_dim=48, depths=[2], num_heads=[6], window_size=8,
                                         mlp_ratio=2., upscale=4, img_range=1., upsampler="nearest+conv", drop_path_rate=0.),
                          (1, 3, 16, 16), 26, True),
    "rdstsr_n_mlp": ("rdstsr_n", dict(img_size=16, in_chans=1, sr_scale=2, embed_dim=48, dense_layer_depths=[2, 2],
                                      num_heads=[6, 6], window_size=[8, 8], rdb_depths=[2, 2], mlp_ratio=2.,
                                      growth_rate=24, pre_norm=True, global_bottleneck=True, global_bottleneck_ratio=1.,
                                      global_bottleneck_mode="mlp", global_res_scale=0.8),
                     (2, 1, 16, 16), 24, True),
    "rdstsr_n_conv": ("rdstsr_n", dict(img_size=16, in_chans=1, sr_scale=2, embed_dim=48, dense_layer_depths=[2, 2],
                                       num_heads=[6, 6], window_size=[8, 8], rdb_depths=[2, 2], mlp_ratio=2.,
                                       growth_rate=24, pre_norm=True, global_bottleneck=True, global_bottleneck_ratio=1.,
                                       global_bottleneck_mode="conv"),
                      (1, 1, 16, 16), 25, True),
    # constructor branches of RDSTSR the factory can select (rdst_variations.py:286-303 'head' mode, :1398-1399 nn.Identity for
    # rdst_layer_norm = False, :1245-1248 / :1330-1331 absolute position embedding; swin_transformer_sr.py:82 qk_scale)
    "rdstsr_head_pre": ("rdstsr", dict(img_size=16, in_chans=1, sr_scale=2, embed_dim=48, dense_layer_depths=[2], num_heads=[6],
                                       window_size=[8], rdb_depths=[2], mlp_ratio=2., growth_rate=24, dim_modify_mode="head",
                                       pre_norm=True, feature_last_operation=True, dense_scale=0.8),
                        (2, 1, 16, 16), 31, True),
    "rdstsr_head_post": ("rdstsr", dict(img_size=16, in_chans=1, sr_scale=2, embed_dim=48, dense_layer_depths=[2], num_heads=[6],
                                        window_size=[8], rdb_depths=[2], mlp_ratio=2., growth_rate=24, dim_modify_mode="head",
                                        pre_norm=False, rdb_residual_scale=0.9),
                         (1, 1, 16, 16), 32, True),
    "rdstsr_identity_norm": ("rdstsr", dict(img_size=16, in_chans=1, sr_scale=2, embed_dim=48, dense_layer_depths=[2], num_heads=[6],
                                            window_size=[8], rdb_depths=[2], mlp_ratio=2., growth_rate=24, pre_norm=True,
                                            norm_layer="identity", feature_last_operation=True),
                             (1, 1, 16, 16), 33, True),
    "rdstsr_ape": ("rdstsr", dict(img_size=16, in_chans=1, sr_scale=2, embed_dim=48, dense_layer_depths=[2], num_heads=[6],
                                  window_size=[8], rdb_depths=[2], mlp_ratio=2., growth_rate=24, pre_norm=True, ape=True,
                                  feature_last_operation=True),
                   (2, 1, 16, 16), 34, True),
    "rdstsr_qk_scale": ("rdstsr", dict(img_size=16, in_chans=1, sr_scale=2, embed_dim=48, dense_layer_depths=[2], num_heads=[6],
                                       window_size=[8], rdb_depths=[2], mlp_ratio=2., growth_rate=24, pre_norm=True, qk_scale=0.3,
                                       feature_last_operation=True),
                        (1, 1, 16, 16), 35, True),
}


def model_kwargs(kw):
    """MODEL_CASES keeps plain data; 'identity' / 'layernorm' name the norm_layer class."""
    import torch.nn as nn
    kw = dict(kw)
    if "norm_layer" in kw:
        kw["norm_layer"] = {"identity": nn.Identity, "layernorm": nn.LayerNorm}[kw["norm_layer"]]
    return kw
