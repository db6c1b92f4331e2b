"""CPU: the oracle reproduces the fixtures captured from the real reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O
from util import NET_CASES, load_golden

BLOCKS = ["block_c60_ws8_s0", "block_c60_ws8_s4", "block_c90_ws8_s4_nonsq", "block_c120_ws8_s4",
          "block_c60_ws16_s8", "block_c48_ws8_clamped"]


@pytest.mark.parametrize("name", BLOCKS)
def test_block_fixture(name):
    g = load_golden(name)
    C, heads, ws, shift, r0, r1, H, W, B = [int(v) for v in g["meta"]]
    sd = {"p." + k[3:]: torch.from_numpy(v).requires_grad_(True) for k, v in g.items() if k.startswith("w::")}
    ews, eshift = O.effective_window((r0, r1), ws, shift)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = O.swin_block(x, (H, W), sd, "p.", heads, ews, eshift)
    y.backward(torch.from_numpy(g["gy"]))
    assert np.abs(y.detach().numpy() - g["y"]).max() <= 1e-5
    assert np.abs(x.grad.numpy() - g["gx"]).max() <= 1e-5
    for k, v in g.items():
        if k.startswith("g::"):
            got = sd["p." + k[3:]].grad.numpy()
            assert np.linalg.norm(got - v) <= 1e-4 * max(np.linalg.norm(v), 1e-12), k


@pytest.mark.parametrize("name", ["net_tiny_64", "net_tiny_b4", "net_e1_16", "net_ws16_32", "net_3conv_x3"])
def test_net_fixture_train(name):
    cfg, seed = NET_CASES[name]
    g = load_golden(name)
    mean = g["mean"].tolist() if "mean" in g else None
    std = g["std"].tolist() if "std" in g else None
    sd = O.make_weights(cfg, seed, mean, std)
    keys = [str(k) for k in g["grad_keys"]]
    sd = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    y = O.rdstsr_forward(torch.from_numpy(g["x"]), sd, cfg)
    tgt = torch.from_numpy(g["target"])
    loss = F.l1_loss(y, tgt)
    loss.backward()
    assert np.abs(y.detach().numpy() - g["y"]).max() <= 2e-5
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    assert abs(O.psnr(tgt, y, border=cfg["sr_scale"]) - float(g["psnr"])) <= 5e-5   # PSNR gate, SURVEY.md §8d
    for k, l2 in zip(keys, g["grad_l2"]):
        assert abs(sd[k].grad.double().norm().item() - l2) <= 2e-4 * max(l2, 1e-9), k
    for k, ref in g.items():      # the gradients the fixture holds elementwise (net_tiny_b4: all of them)
        if k.startswith("grad::"):
            got = sd[k[6:]].grad.numpy()
            assert np.linalg.norm(got - ref) <= 2e-4 * max(np.linalg.norm(ref), 1e-12), k


def test_net_fixture_eval_nonsquare():
    cfg, seed = NET_CASES["net_e1_eval_40x32"]
    g = load_golden("net_e1_eval_40x32")
    with torch.no_grad():
        y = O.rdstsr_forward(torch.from_numpy(g["x"]), O.make_weights(cfg, seed), cfg)
    assert y.shape == (1, 1, 160, 128)
    assert np.abs(y.numpy() - g["y"]).max() <= 2e-5


def test_mask_and_index_restatements():
    # analytic region rule used by the HIP kernels == the slice construction of the reference
    for (H, W, ws, s) in [(16, 16, 8, 4), (16, 24, 8, 4), (8, 8, 8, 4), (32, 32, 16, 8), (8, 12, 4, 2)]:
        m = O.calculate_mask(H, W, ws, s)
        rid = lambda r, n: 0 if r < n - ws else (1 if r < n - s else 2)  # noqa: E731
        for wi in range(m.shape[0]):
            wr, wc = divmod(wi, W // ws)
            reg = [3 * rid(wr * ws + t // ws, H) + rid(wc * ws + t % ws, W) for t in range(ws * ws)]
            reg = torch.tensor(reg)
            exp = torch.where(reg[None, :] != reg[:, None], -100.0, 0.0)
            assert torch.equal(m[wi], exp)
    idx = O.relative_position_index(8)
    assert idx.shape == (64, 64) and idx.min() == 0 and idx.max() == 224 and idx[0, 0] == 112
