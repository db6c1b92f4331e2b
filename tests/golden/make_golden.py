"""Generate the committed golden fixtures from the REAL reference (build container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; CPU only)

What it does, per case:
  1. imports GinZhu/RDST's own network code from /root/reference (with a 3-symbol stub
     for ``timm.models.layers`` — DropPath is never active on this path);
  2. checks that the reference's ``state_dict()`` keys / shapes / dtypes equal
     ``oracle.rdst_oracle.state_dict_layout`` and loads ``make_weights`` into it (strict);
  3. runs the reference forward (+ L1-loss backward) on seeded inputs, runs the oracle on
     the same inputs and asserts they agree (this is the oracle's pin);
  4. writes inputs + reference outputs / loss / gradients into ``tests/golden/*.npz`` and
     the state-dict layout into ``tests/golden/state_dict_*.json``.

Only DATA is written: no reference source travels.  The GPU box never runs this script.
"""
from __future__ import annotations

import json
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import rdst_oracle as O  # noqa: E402


def _import_reference():
    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training, "stochastic depth is only the identity in eval()"
            return x

    layers = types.ModuleType("timm.models.layers")
    layers.DropPath = DropPath
    layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    layers.trunc_normal_ = nn.init.trunc_normal_
    sys.modules.setdefault("timm", types.ModuleType("timm"))
    sys.modules.setdefault("timm.models", types.ModuleType("timm.models"))
    sys.modules["timm.models.layers"] = layers
    # The reference's `networks` is a namespace package (no __init__.py); this repository's drop-in shim
    # `networks/` is a regular package and would win regardless of path order, so hide the repo root while
    # the reference is imported.
    saved = list(sys.path)
    sys.path[:] = [REF] + [q for q in saved if os.path.abspath(q or ".") != ROOT]
    for k in [k for k in sys.modules if k == "networks" or k.startswith("networks.")]:
        del sys.modules[k]
    import networks.rdst_variations as rv  # noqa
    import networks.swin_transformer_sr as st  # noqa
    assert rv.__file__.startswith(REF) and st.__file__.startswith(REF), (rv.__file__, st.__file__)
    sys.path[:] = saved
    return rv, st


def seeded(shape, seed, lo=0.0, hi=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.uniform(lo, hi, size=shape).astype(np.float32))


def seeded_normal(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))


def build_ref(rv, cfg, mean=None, std=None):
    net = rv.RDSTSR(
        img_size=cfg["img_size"], patch_size=1, in_chans=cfg["in_chans"], sr_scale=cfg["sr_scale"],
        embed_dim=cfg["embed_dim"], dense_layer_depths=cfg["dense_layer_depths"],
        num_heads=cfg["num_heads"], window_size=cfg["window_size"], rdb_depths=cfg["rdb_depths"],
        mlp_ratio=cfg["mlp_ratio"], qk_scale=cfg["qk_scale"],
        norm_layer=nn.LayerNorm if cfg["layer_norm"] else nn.Identity, patch_norm=cfg["patch_norm"],
        resi_connection=cfg["resi_connection"], growth_rate=cfg["growth_rate"],
        dense_scale=cfg["dense_scale"], rdb_residual_scale=cfg["rdb_residual_scale"],
        global_res_scale=cfg["global_res_scale"], mean=mean, std=std, pre_norm=cfg["pre_norm"],
        feature_last_operation=cfg["feature_last_operation"])
    return net


def check_layout(net, cfg, name):
    ref_sd = net.state_dict()
    layout = O.state_dict_layout(cfg)
    ref_layout = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()}
    mine = {k: [list(s), dt] for k, (s, dt, _kind) in layout.items()}
    assert list(ref_layout.keys()) == list(mine.keys()), "state-dict key order/name mismatch"
    assert ref_layout == mine, "state-dict shape/dtype mismatch"
    n_params = sum(p.numel() for p in net.parameters())
    n_train = sum(p.numel() for p in net.parameters() if p.requires_grad)
    with open(os.path.join(HERE, f"state_dict_{name}.json"), "w") as f:
        json.dump({"entries": ref_layout, "n_params": n_params, "n_trainable": n_train,
                   "trainable": [k for k, p in net.named_parameters() if p.requires_grad]}, f)
    print(f"[{name}] layout ok: {len(ref_layout)} entries, {n_params} params ({n_train} trainable)")


ONLY = None   # `python make_golden.py --only NAME [NAME ...]`: regenerate just these fixtures


def run_net_case(rv, name, cfg, x, seed, mean=None, std=None, train=True, grad_keys=(), adam_keys=None):
    """Full-network case: reference fwd (+bwd of L1 vs seeded target).  grad_keys = "all": every gradient elementwise."""
    if ONLY is not None and name not in ONLY:
        return
    net = build_ref(rv, cfg, mean, std)
    check_layout(net, cfg, name)
    sd = O.make_weights(cfg, seed, mean, std)
    net.load_state_dict(sd, strict=True)
    # buffers the reference built itself must equal the oracle's restatement
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k]), k
    out = {"x": x.numpy(), "seed": np.int64(seed)}
    if mean is not None:
        out["mean"] = np.asarray(mean, np.float32)
        out["std"] = np.asarray(std, np.float32)
    if train:
        net.train()
        y = net(x)
        tgt = seeded(tuple(y.shape), seed + 1000)
        loss = F.l1_loss(y, tgt)
        loss.backward()
        # parameters the forward never touches (e.g. conv_after_body when feature_last_operation is
        # False) keep grad None in the reference; they are recorded as absent.
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()
                 if p.requires_grad and p.grad is not None}
        # oracle on the same inputs (fp32), with autograd
        osd = {k: (v.clone().requires_grad_(True) if k in grads else v) for k, v in sd.items()}
        oy = O.rdstsr_forward(x, osd, cfg)
        ol = F.l1_loss(oy, tgt)
        ol.backward()
        err = (oy - y).abs().max().item()
        assert err <= 2e-5, f"oracle fwd mismatch {err}"
        assert abs(ol.item() - loss.item()) <= 1e-6
        worst = 0.0
        for k, g in grads.items():
            og = osd[k].grad
            rel = (og - g).norm().item() / max(g.norm().item(), 1e-12)
            worst = max(worst, rel)
            assert rel <= 2e-4, f"oracle grad mismatch {k}: {rel}"
        print(f"[{name}] oracle==reference: fwd max|d|={err:.2e}, worst grad rel L2={worst:.2e}")
        out.update(y=y.detach().numpy(), target=tgt.numpy(), loss=np.float64(loss.item()),
                   psnr=np.float64(O.psnr(tgt, y, border=cfg["sr_scale"])))
        keys = sorted(grads)
        out["grad_keys"] = np.array(keys)
        out["grad_l2"] = np.array([grads[k].double().norm().item() for k in keys])
        out["grad_sum"] = np.array([grads[k].double().sum().item() for k in keys])
        if grad_keys == "all":
            grad_keys = keys
        for k in grad_keys:
            out["grad::" + k] = grads[k].numpy()
        # parameters after ONE Adam step with the ini's hyper-parameters
        # (config_files/RDST_E1_OASIS_example_SRx4.ini:128-135; utils/optim.py:30-53)
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4,
                               betas=(0.9, 0.99), eps=1e-8, weight_decay=0)
        opt.step()
        for k in (grad_keys if adam_keys is None else adam_keys):
            out["adam1::" + k] = dict(net.named_parameters())[k].detach().numpy()
    else:
        net.eval()
        with torch.no_grad():
            y = net(x)
            oy = O.rdstsr_forward(x, sd, cfg)
        err = (oy - y).abs().max().item()
        assert err <= 2e-5, f"oracle fwd mismatch {err}"
        print(f"[{name}] oracle==reference (eval): fwd max|d|={err:.2e}")
        out["y"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)


def run_block_case(st, name, C, heads, ws, shift, res, x_size, B, seed):
    """One reference SwinTransformerBlock fwd+bwd (pins norm1/qkv/attention/proj/mlp)."""
    if ONLY is not None and name not in ONLY:
        return
    blk = st.SwinTransformerBlock(dim=C, input_resolution=res, num_heads=heads, window_size=ws,
                                  shift_size=shift, mlp_ratio=2.0)
    rng_sd = {}
    for k, v in blk.state_dict().items():
        if k in ("attn_mask", "attn.relative_position_index"):
            rng_sd[k] = v
            continue
        n = seeded_normal(tuple(v.shape), [seed, zlib.crc32(k.encode())])
        if k.endswith("bias_table"):
            rng_sd[k] = 0.5 * n
        elif k.startswith("norm") and k.endswith("weight"):
            rng_sd[k] = 1.0 + 0.1 * n
        elif k.endswith("bias"):
            rng_sd[k] = 0.05 * n
        else:
            rng_sd[k] = 0.7 * n / np.sqrt(v.shape[1])
    blk.load_state_dict(rng_sd, strict=True)
    H, W = x_size
    x = seeded_normal((B, H * W, C), seed + 1).requires_grad_(True)
    gy = seeded_normal((B, H * W, C), seed + 2)
    y = blk(x, x_size)
    y.backward(gy)
    ews, eshift = O.effective_window(res, ws, shift)
    osd = {"p." + k: (v.clone().requires_grad_(v.dtype.is_floating_point and k != "attn_mask"))
           for k, v in rng_sd.items()}
    ox = x.detach().clone().requires_grad_(True)
    oy = O.swin_block(ox, x_size, osd, "p.", heads, ews, eshift)
    oy.backward(gy)
    assert (oy - y).abs().max().item() <= 1e-5
    assert (ox.grad - x.grad).abs().max().item() <= 1e-5
    out = {"x": x.detach().numpy(), "gy": gy.numpy(), "y": y.detach().numpy(), "gx": x.grad.numpy(),
           "meta": np.array([C, heads, ws, shift, res[0], res[1], H, W, B], np.int64)}
    for k, p in blk.named_parameters():
        rel = (osd["p." + k].grad - p.grad).norm().item() / max(p.grad.norm().item(), 1e-12)
        assert rel <= 1e-4, (k, rel)
        out["w::" + k] = p.detach().numpy()
        out["g::" + k] = p.grad.numpy()
    print(f"[{name}] block oracle==reference")
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)


def run_model_case(rv, st, name):
    """'Next'-row models (SwinIR baseline, RDSTSR_N): fixtures straight from the reference module."""
    if ONLY is not None and name not in ONLY:
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import MODEL_CASES, model_kwargs, seeded_fill
    kind, kw, xshape, seed, train = MODEL_CASES[name]
    net = {"swinir": st.SwinIR, "rdstsr_n": rv.RDSTSR_N, "rdstsr": rv.RDSTSR}[kind](**model_kwargs(kw))
    net.load_state_dict(seeded_fill(net.state_dict(), seed), strict=True)
    layout = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, f"state_dict_{name}.json"), "w") as f:
        json.dump({"entries": layout}, f)
    x = seeded(xshape, seed + 500)
    out = {"x": x.numpy()}
    if train:
        net.train()
        y = net(x)
        tgt = seeded(tuple(y.shape), seed + 1000)
        loss = F.l1_loss(y, tgt)
        loss.backward()
        grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
        keys = sorted(grads)
        out.update(y=y.detach().numpy(), target=tgt.numpy(), loss=np.float64(loss.item()), grad_keys=np.array(keys),
                   grad_l2=np.array([grads[k].double().norm().item() for k in keys]))
        for k in keys:          # EVERY gradient elementwise (a permuted or sign-flipped gradient of equal norm must not pass)
            out["grad::" + k] = grads[k].numpy()
    else:
        net.eval()
        with torch.no_grad():
            out["y"] = net(x).numpy()
    print(f"[{name}] fixture from the reference: out {tuple(out['y'].shape)}, {len(layout)} state-dict entries")
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)


def main():
    global ONLY
    if "--only" in sys.argv:
        ONLY = set(sys.argv[sys.argv.index("--only") + 1:])
    torch.manual_seed(0)
    torch.set_num_threads(8)
    rv, st = _import_reference()

    # --- block-level fixtures (C, heads, ws, shift, ctor resolution, run size, batch) -----
    run_block_case(st, "block_c60_ws8_s0", 60, 6, 8, 0, (16, 16), (16, 16), 2, 11)
    run_block_case(st, "block_c60_ws8_s4", 60, 6, 8, 4, (16, 16), (16, 16), 2, 12)
    run_block_case(st, "block_c90_ws8_s4_nonsq", 90, 6, 8, 4, (64, 64), (16, 24), 1, 13)  # on-the-fly mask
    run_block_case(st, "block_c120_ws8_s4", 120, 6, 8, 4, (24, 24), (24, 24), 1, 14)
    run_block_case(st, "block_c60_ws16_s8", 60, 6, 16, 8, (32, 32), (32, 32), 1, 15)
    run_block_case(st, "block_c48_ws8_clamped", 48, 6, 8, 4, (8, 8), (8, 8), 2, 16)  # ws clamp -> shift 0

    # --- network-level fixtures -----------------------------------------------------------
    gk_tiny = ["head.weight", "body.0.body.0.body.blocks.1.attn.relative_position_bias_table",
               "body.1.body.2.body.blocks.0.attn.qkv.weight", "body.0.conv.weight", "tail.1.weight",
               "body.1.body.1.tail.1.weight", "norm.weight"]
    run_net_case(rv, "net_tiny_64", O.CFG_TINY, seeded((1, 1, 64, 64), 101), 1, grad_keys=gk_tiny)
    # BASELINE.json configs[0] at its stated shape (4 x 1 x 64 x 64), EVERY gradient elementwise
    run_net_case(rv, "net_tiny_b4", O.CFG_TINY, seeded((4, 1, 64, 64), 106), 1, grad_keys="all", adam_keys=gk_tiny)
    gk_e1 = ["head.weight", "body.7.body.2.body.blocks.1.attn.relative_position_bias_table",
             "body.3.body.1.body.blocks.0.mlp.fc1.weight", "body.0.conv.bias", "tail.0.2.weight"]
    run_net_case(rv, "net_e1_16", O.CFG_E1, seeded((1, 1, 16, 16), 102), 2, grad_keys=gk_e1)
    # whole-slice inference shape of the tester: non-square, != ctor img_size
    run_net_case(rv, "net_e1_eval_40x32", O.CFG_E1, seeded((1, 1, 40, 32), 103), 2, train=False)
    # mean/std shift, 3 channels, window 16, x2 (BASELINE.json configs[3], shortened to 2 blocks)
    cfg16 = O.make_cfg(**{**O.CFG_WS16, "img_size": 32, "dense_layer_depths": [2, 2], "num_heads": [6, 6],
                          "window_size": [16, 16], "rdb_depths": [3, 2]})
    run_net_case(rv, "net_ws16_32", cfg16, seeded((1, 3, 32, 32), 104), 3,
                 mean=[0.45, 0.40, 0.50], std=[0.9, 1.1, 1.0],
                 grad_keys=["head.weight", "body.1.conv.weight"])
    # '3conv' residual connection (LeakyReLU variant) + post-norm tail + no feature_last_operation
    cfg3 = O.make_cfg(img_size=16, in_chans=1, sr_scale=3, embed_dim=48, dense_layer_depths=[2], num_heads=[6],
                      window_size=[8], rdb_depths=[2], mlp_ratio=2.0, growth_rate=24, pre_norm=False,
                      resi_connection="3conv", feature_last_operation=False, dense_scale=0.5,
                      rdb_residual_scale=0.7, global_res_scale=0.9)
    run_net_case(rv, "net_3conv_x3", cfg3, seeded((2, 1, 16, 16), 105), 4,
                 grad_keys=["body.0.conv.0.weight", "body.0.body.0.tail.0.weight"])
    # --- "next" rows: SwinIR baseline and the RDSTSR_N bottleneck variant ----------------------------
    for name in ("swinir_ps_x4", "swinir_psd_x2_rgb", "swinir_denoise", "swinir_nearest_x4", "rdstsr_n_mlp", "rdstsr_n_conv"):
        run_model_case(rv, st, name)
    # --- constructor branches of RDSTSR that make_RDSTSR can select ('head' dim modifier pre- / post-norm, nn.Identity
    # norms, absolute position embedding, qk_scale): fixtures straight from the reference, every gradient elementwise
    for name in ("rdstsr_head_pre", "rdstsr_head_post", "rdstsr_identity_norm", "rdstsr_ape", "rdstsr_qk_scale"):
        run_model_case(rv, st, name)


if __name__ == "__main__":
    main()
