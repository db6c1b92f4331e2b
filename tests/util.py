"""Shared helpers for the test-suite."""
import os

import numpy as np
import torch

from oracle import rdst_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

NET_CASES = {
    "net_tiny_64": (O.CFG_TINY, 1),
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


def build_net(cfg, mean=None, std=None):
    """The HIP-backed RDSTSR for an oracle cfg dict (same kwargs the golden generator gave the reference)."""
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
        feature_last_operation=cfg["feature_last_operation"])
