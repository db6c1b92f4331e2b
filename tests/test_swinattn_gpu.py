"""K8 (csrc/swinattn_fwd.hip): LayerNorm + qkv -> window attention -> proj + shortcut in ONE launch, through the C ABI.

(1) against the three-call sequence it replaces (rdst_ln_linear_fwd -> rdst_wattn_fwd -> rdst_ln_linear_fwd): the fused kernel
    runs the same arithmetic operation for operation, so qkv, the attention output, x1 and the LayerNorm statistics must be
    BIT-IDENTICAL — every width, shifted and unshifted, window counts below / at / above the 256 workgroups of a launch (one
    window per workgroup, ragged last round, 8 rounds), non-square images, x as a strided slice of a wider (dense) buffer;
(2) against the CPU oracle's restatement of networks/swin_transformer_sr.py:240-271 (fp32) within the bf16 bound."""
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O
from util import rand

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(C, seed):
    return dict(n1w=1 + 0.1 * rand((C,), seed + 1), n1b=0.1 * rand((C,), seed + 2), qkvw=rand((3 * C, C), seed + 3, C ** -0.5),
                qkvb=0.1 * rand((3 * C,), seed + 4), table=0.5 * rand((225, 6), seed + 5), projw=rand((C, C), seed + 6, C ** -0.5),
                projb=0.1 * rand((C,), seed + 7))


def _fused(lib, _lib, x, ldx, P, B, H, W, C, shift, scale, out=None, wsp=None):
    M = B * H * W
    if out is None:
        qkv = torch.full((M, 3 * C), float("nan"), dtype=torch.bfloat16, device=DEV)
        a = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        x1 = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        stats = torch.full((M, 2), float("nan"), device=DEV)
    else:
        qkv, a, x1, stats = out
    nws = lib.rdst_swin_attn_fwd_workspace(C)
    if wsp is None:
        wsp = torch.empty(nws, dtype=torch.uint8, device=DEV)
    else:
        nws = _lib.PREPACKED
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.rdst_swin_attn_fwd(x.data_ptr(), ldx, P["n1w"].data_ptr(), P["n1b"].data_ptr(), P["qkvw"].data_ptr(),
                                      P["qkvb"].data_ptr() if P["qkvb"] is not None else None, P["table"].data_ptr(),
                                      P["projw"].data_ptr(), P["projb"].data_ptr(), qkv.data_ptr(), 3 * C, a.data_ptr(), C,
                                      x1.data_ptr(), C, stats.data_ptr(), wsp.data_ptr(), nws, B, H, W, C, 6, 8, shift, scale,
                                      _lib.BF16, st), "rdst_swin_attn_fwd")
    return qkv, a, x1, stats


def _unfused(lib, _lib, x, ldx, P, B, H, W, C, shift, scale, out=None, wsp=None):
    M = B * H * W
    if out is None:
        qkv = torch.empty((M, 3 * C), dtype=torch.bfloat16, device=DEV)
        a = torch.empty((M, C), dtype=torch.bfloat16, device=DEV)
        x1 = torch.empty((M, C), dtype=torch.bfloat16, device=DEV)
        stats = torch.empty((M, 2), device=DEV)
    else:
        qkv, a, x1, stats = out
    st = torch.cuda.current_stream().cuda_stream
    n1 = lib.rdst_ln_linear_fwd_workspace(C, 3 * C)
    n2 = lib.rdst_ln_linear_fwd_workspace(C, C)
    if wsp is None:
        w1 = torch.empty(n1, dtype=torch.uint8, device=DEV)
        w2 = torch.empty(n2, dtype=torch.uint8, device=DEV)
    else:
        w1, w2 = wsp
        n1 = n2 = _lib.PREPACKED
    _lib.check(lib.rdst_ln_linear_fwd(x.data_ptr(), ldx, P["n1w"].data_ptr(), P["n1b"].data_ptr(), 0, P["qkvw"].data_ptr(),
                                      P["qkvb"].data_ptr() if P["qkvb"] is not None else None, None, 0, qkv.data_ptr(), 3 * C,
                                      stats.data_ptr(), w1.data_ptr(), n1, M, C, 3 * C, 1.0, _lib.BF16, st), "qkv")
    _lib.check(lib.rdst_wattn_fwd(qkv.data_ptr(), 3 * C, P["table"].data_ptr(), None, 0, a.data_ptr(), C, B, H, W, C, 6, 8, shift,
                                  scale, _lib.BF16, st), "wattn")
    _lib.check(lib.rdst_ln_linear_fwd(a.data_ptr(), C, None, None, 0, P["projw"].data_ptr(), P["projb"].data_ptr(), x.data_ptr(), ldx,
                                      x1.data_ptr(), C, None, w2.data_ptr(), n2, M, C, C, 1.0, _lib.BF16, st), "proj")
    return qkv, a, x1, stats


GEOMS = [(1, 8, 8), (2, 16, 24), (3, 40, 32), (4, 64, 64), (5, 64, 64), (9, 56, 64)]   # 1 / 12 / 60 / 256 / 320 / 504 windows


@pytest.mark.parametrize("C", [60, 90, 120])
@pytest.mark.parametrize("shift", [0, 4, 3])
@pytest.mark.parametrize("B,H,W", GEOMS)
def test_fused_equals_the_three_calls_bit_for_bit(C, shift, B, H, W):
    from rdst_amd import _lib
    lib = _lib.load()
    assert lib.rdst_swin_attn_fwd_supported(C, 6, 8, _lib.BF16) == 1
    M = B * H * W
    scale = (C // 6) ** -0.5
    P = {k: v.to(DEV).contiguous() for k, v in _params(C, 10 * C + shift).items()}
    # x = the first C channels of a 150-wide dense buffer (ld = 150): what a DenseSTLayer's first block reads
    wide = rand((M, 150), 99 + C).to(DEV).bfloat16()
    x = wide[:, :C]
    got = _fused(lib, _lib, x, 150, P, B, H, W, C, shift, scale)
    want = _unfused(lib, _lib, x, 150, P, B, H, W, C, shift, scale)
    torch.cuda.synchronize()
    for name, g, w in zip(("qkv", "a", "x1", "stats"), got, want):
        assert torch.isfinite(g.float()).all(), name
        assert torch.equal(g, w), (name, (g.float() - w.float()).abs().max().item())


@pytest.mark.parametrize("C", [60, 90, 120])
def test_fused_full_size_and_no_qkv_bias(C):
    """The benchmark geometry (32 x 64 x 64: 2048 windows, 8 per workgroup), contiguous x, qkv_bias=False."""
    from rdst_amd import _lib
    lib = _lib.load()
    B, H, W = 32, 64, 64
    M = B * H * W
    scale = (C // 6) ** -0.5
    P = {k: v.to(DEV).contiguous() for k, v in _params(C, 3 * C).items()}
    P["qkvb"] = None
    x = rand((M, C), 7 + C).to(DEV).bfloat16()
    for shift in (0, 4):
        got = _fused(lib, _lib, x, C, P, B, H, W, C, shift, scale)
        want = _unfused(lib, _lib, x, C, P, B, H, W, C, shift, scale)
        torch.cuda.synchronize()
        for name, g, w in zip(("qkv", "a", "x1", "stats"), got, want):
            assert torch.equal(g, w), (name, shift)


@pytest.mark.parametrize("C", [60, 90, 120])
@pytest.mark.parametrize("shift", [0, 4])
def test_fused_vs_oracle(C, shift):
    """Against the oracle's fp32 restatement of the attention half of SwinTransformerBlock.forward
    (networks/swin_transformer_sr.py:240-271) on the same bf16-representable inputs; tolerance = the bf16 bound of the
    other bf16 kernel tests (intermediates qkv and a are rounded to bf16 where the reference keeps fp32)."""
    from rdst_amd import _lib
    lib = _lib.load()
    B, H, W = 3, 24, 32
    M = B * H * W
    scale = (C // 6) ** -0.5
    Pc = _params(C, 77 + C + shift)
    x = rand((M, C), 5 + C).bfloat16().float()
    h = O.layer_norm(x, Pc["n1w"], Pc["n1b"])
    qkv = F.linear(h, Pc["qkvw"], Pc["qkvb"])
    a = O.window_attention_core(qkv.reshape(B, H, W, 3 * C), Pc["table"], 6, 8, shift, scale).reshape(M, C)
    x1 = x + F.linear(a, Pc["projw"], Pc["projb"])
    P = {k: v.to(DEV).contiguous() for k, v in Pc.items()}
    got = _fused(lib, _lib, x.to(DEV).bfloat16(), C, P, B, H, W, C, shift, scale)
    torch.cuda.synchronize()
    rel = lambda g, w: (g.float().cpu() - w).norm().item() / w.norm().item()
    assert rel(got[0], qkv) <= 8e-3 and rel(got[1], a) <= 1.2e-2 and rel(got[2], x1) <= 8e-3, (rel(got[0], qkv), rel(got[1], a), rel(got[2], x1))
    want_stats = torch.stack([x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()], dim=1)
    torch.testing.assert_close(got[3].cpu(), want_stats, rtol=2e-5, atol=2e-6)


def test_unsupported_shapes_say_so_and_the_block_falls_back():
    from rdst_amd import _lib, ops
    lib = _lib.load()
    assert lib.rdst_swin_attn_fwd_supported(48, 6, 8, _lib.BF16) == 0
    assert lib.rdst_swin_attn_fwd_supported(120, 6, 16, _lib.BF16) == 0
    assert lib.rdst_swin_attn_fwd_supported(120, 6, 8, _lib.F32) == 0
    # a width the fused kernel does not cover runs the three calls; same autograd node either way
    C = 48
    P = {k: v.to(DEV) for k, v in _params(C, 3).items()}
    x = rand((2, 16, 16, C), 4).to(DEV).bfloat16().requires_grad_(True)
    fc1w, fc1b = rand((2 * C, C), 8, C ** -0.5).to(DEV), torch.zeros(2 * C, device=DEV)
    fc2w, fc2b = rand((C, 2 * C), 9, (2 * C) ** -0.5).to(DEV), torch.zeros(C, device=DEV)
    y = ops.swin_block(x, P["n1w"], P["n1b"], P["qkvw"], P["qkvb"], P["table"], P["projw"], P["projb"], P["n1w"], P["n1b"], fc1w, fc1b,
                       fc2w, fc2b, 16, 16, 6, 8, 4, (C // 6) ** -0.5)
    y.float().sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and torch.isfinite(x.grad.float()).all()


@pytest.mark.parametrize("C", [60, 120])
def test_block_through_autograd_fused_on_and_off_agree(C, monkeypatch):
    """ops.swin_block with K8 on and off: identical outputs and identical gradients (the backward reads the same saved tensors)."""
    from rdst_amd import ops
    B, H, W = 2, 32, 24
    P = {k: v.to(DEV).requires_grad_(True) for k, v in _params(C, 21).items()}
    n2w, n2b = (1 + 0.1 * rand((C,), 31)).to(DEV).requires_grad_(True), (0.1 * rand((C,), 32)).to(DEV).requires_grad_(True)
    fc1w, fc1b = rand((2 * C, C), 33, C ** -0.5).to(DEV).requires_grad_(True), (0.1 * rand((2 * C,), 34)).to(DEV).requires_grad_(True)
    fc2w, fc2b = rand((C, 2 * C), 35, (2 * C) ** -0.5).to(DEV).requires_grad_(True), (0.1 * rand((C,), 36)).to(DEV).requires_grad_(True)
    leaves = [P[k] for k in ("n1w", "n1b", "qkvw", "qkvb", "table", "projw", "projb")] + [n2w, n2b, fc1w, fc1b, fc2w, fc2b]
    x0 = rand((B, H, W, C), 40).to(DEV).bfloat16()
    gy = rand((B, H, W, C), 41).to(DEV).bfloat16()
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "ATTN_FUSED", on)
        x = x0.clone().requires_grad_(True)
        for t in leaves:
            t.grad = None
        y = ops.swin_block(x, *leaves, H, W, 6, 8, 4, (C // 6) ** -0.5)
        y.backward(gy)
        torch.cuda.synchronize()
        res[on] = [y.detach().clone(), x.grad.clone()] + [t.grad.clone() for t in leaves]
    for i, (u, v) in enumerate(zip(res[True], res[False])):
        assert torch.equal(u, v), i


def test_network_eval_nonsquare_through_k8_equals_the_three_call_path(monkeypatch):
    """The whole RDST-E1 network in eval mode on a non-square whole slice (40 x 72: 45 windows per image, not a multiple of
    anything), bf16: with K8 and with the three-call composition the SR image is bit-identical (K8 runs the same arithmetic),
    and so is a training step's loss and every parameter gradient."""
    import torch.nn.functional as F2
    from rdst_amd import ops
    from util import build_net
    cfg = O.CFG_E1
    sd = O.make_weights(cfg, 5)
    x = rand((2, 1, 40, 72), 3).abs().clamp(0, 1)
    tgt = rand((2, 1, 160, 288), 4).abs().clamp(0, 1)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "ATTN_FUSED", on)
        net = build_net(cfg)
        net.load_state_dict(sd, strict=True)
        net.to(DEV).set_compute_dtype(torch.bfloat16)
        net.eval()
        with torch.no_grad():
            y_eval = net(x.to(DEV)).clone()
        net.train()
        loss = F2.l1_loss(net(x.to(DEV)), tgt.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        res[on] = (y_eval, loss.detach().clone(), [p.grad.clone() for p in net.parameters() if p.requires_grad])
    assert torch.isfinite(res[True][0]).all()
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    assert len(res[True][2]) == 750 and all(torch.equal(a, b) for a, b in zip(res[True][2], res[False][2]))
