"""Dropout p > 0 (training mode): swin_drop_rate / swin_attn_drop_rate of make_RDSTSR reach nn.Dropout in four places of the
reference — on the attention weights (swin_transformer_sr.py:136), behind proj (:140), twice inside the Mlp (:26, :28) and on
the embedded patches (rdst_variations.py:1332).  The random stream of torch's CPU generator is not reproducible on another
device, so parity is stated GIVEN THE MASK: the attention kernels' counter-based mask is exported (rdst_wattn_drop_mask) and
handed to the oracle; the elementwise dropouts are torch ops on the device tensors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))


def _export_mask(B, H, W, heads, ws, p, seed):
    from rdst_amd import _lib
    lib = _lib.load()
    nblk, N = B * (H // ws) * (W // ws) * heads, ws * ws
    m = torch.empty(nblk, N, N, dtype=torch.float32, device=DEV)
    _lib.check(lib.rdst_wattn_drop_mask(m.data_ptr(), B, H, W, heads, ws, float(p), seed.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "rdst_wattn_drop_mask")
    torch.cuda.synchronize()
    return m.cpu().view(-1, heads, N, N)


CASES = [
    # B, H, W, C, heads, ws, shift, p
    (2, 16, 16, 60, 6, 8, 0, 0.1),
    (2, 16, 16, 60, 6, 8, 4, 0.25),
    (1, 16, 24, 90, 6, 8, 4, 0.5),
    (1, 32, 32, 60, 6, 16, 8, 0.1),     # window 16: the MFMA kernels are bypassed too
    (2, 8, 12, 72, 6, 4, 2, 0.3),
]


@pytest.mark.parametrize("B,H,W,C,heads,ws,shift,p", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_dropout_given_the_mask(B, H, W, C, heads, ws, shift, p, dtype):
    from rdst_amd import ops
    scale = (C // heads) ** -0.5
    qkv = _rand((B, H, W, 3 * C), 31).to(dtype)
    table = _rand(((2 * ws - 1) ** 2, heads), 32, 0.5)
    gout = _rand((B, H, W, C), 33).to(dtype)
    seed = torch.tensor([123456789 + ws], dtype=torch.int64, device=DEV)

    q = qkv.to(DEV).requires_grad_(True)
    t = table.to(DEV).requires_grad_(True)
    o = ops.window_attention(q, t, H, W, heads, ws, shift, scale, attn_drop=p, seed=seed)
    o.backward(gout.to(DEV))
    torch.cuda.synchronize()

    M = _export_mask(B, H, W, heads, ws, p, seed)
    keep = 1.0 / (1.0 - p)
    vals = torch.unique(M).tolist()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - keep) <= 1e-6 * keep, vals
    frac = (M > 0).double().mean().item()
    n = M.numel()
    assert abs(frac - (1 - p)) <= 5 * (p * (1 - p) / n) ** 0.5 + 1e-4, frac          # Bernoulli(1 - p) within 5 sigma

    q_ref = qkv.float().clone().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale, drop_mult=M)
    o_ref.backward(gout.float())

    def rel(a, b):
        return (a.float().cpu().double() - b.double()).norm().item() / b.double().norm().item()

    tol = 3e-6 if dtype == torch.float32 else 2e-2
    assert rel(o, o_ref) <= tol, rel(o, o_ref)
    assert rel(q.grad, q_ref.grad) <= tol, rel(q.grad, q_ref.grad)
    assert rel(t.grad, t_ref.grad) <= tol, rel(t.grad, t_ref.grad)
    for s3 in range(3):
        a, b_ = q.grad[..., s3 * C:(s3 + 1) * C], q_ref.grad[..., s3 * C:(s3 + 1) * C]
        assert rel(a, b_) <= 1.5 * tol, (s3, rel(a, b_))
    # and it is not the un-dropped result
    o0 = O.window_attention_core(qkv.float(), table, heads, ws, shift, scale)
    assert rel(o, o0) > 10 * tol


def test_mask_depends_on_the_seed_only():
    a = _export_mask(1, 16, 16, 6, 8, 0.3, torch.tensor([1], dtype=torch.int64, device=DEV))
    b = _export_mask(1, 16, 16, 6, 8, 0.3, torch.tensor([1], dtype=torch.int64, device=DEV))
    c = _export_mask(1, 16, 16, 6, 8, 0.3, torch.tensor([2], dtype=torch.int64, device=DEV))
    assert torch.equal(a, b) and not torch.equal(a, c)
    # windows / heads / rows are decorrelated: the keep rate of every (window, head) block is near 1 - p
    per = (a > 0).float().mean(dim=(2, 3))
    assert (per - 0.7).abs().max().item() < 0.05


def test_bad_probability_is_refused():
    from rdst_amd import _lib, ops
    from rdst_amd.networks.swin_transformer_sr import Mlp, WindowAttention
    with pytest.raises(ValueError):
        Mlp(60, 120, drop=1.0)
    with pytest.raises(ValueError):
        WindowAttention(60, (8, 8), 6, attn_drop=-0.1)
    qkv = torch.zeros(1, 8, 8, 180, device=DEV)
    with pytest.raises(ValueError):
        ops.window_attention(qkv, torch.zeros(225, 6, device=DEV), 8, 8, 6, 8, 0, 1.0, attn_drop=1.0)
    lib = _lib.load()
    out = torch.empty(1, 8, 8, 60, device=DEV)
    rc = lib.rdst_wattn_fwd_drop(qkv.data_ptr(), 180, torch.zeros(225, 6, device=DEV).data_ptr(), None, 0, out.data_ptr(), 60,
                                 1, 8, 8, 60, 6, 8, 0, 1.0, _lib.F32, 0.5, None, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"seed" in lib.rdst_last_error()


def _block(drop, attn_drop, seed=3):
    from rdst_amd.networks.swin_transformer_sr import SwinTransformerBlock
    torch.manual_seed(seed)
    blk = SwinTransformerBlock(60, (16, 16), 6, window_size=8, shift_size=4, mlp_ratio=2.0, drop=drop, attn_drop=attn_drop)
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(torch.randn_like(prm) * 0.1 + (1.0 if prm.dim() == 1 and prm.shape[0] == 60 else 0.0))
    return blk.to(DEV)


def test_block_eval_ignores_dropout_and_training_is_seeded():
    x = _rand((2, 256, 60), 41).to(DEV)
    a, b = _block(0.0, 0.0), _block(0.2, 0.2)
    b.load_state_dict(a.state_dict())
    a.eval(), b.eval()
    with torch.no_grad():
        assert torch.equal(a(x, (16, 16)), b(x, (16, 16)))          # nn.Dropout is the identity in eval(): the fused path both times
    b.train()
    assert not b.fuses_input_gradient()
    outs = []
    for s in (7, 7, 8):
        torch.manual_seed(s)
        xi = x.clone().requires_grad_(True)
        for prm in b.parameters():
            prm.grad = None
        y = b(xi, (16, 16))
        y.square().mean().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), xi.grad.clone(), [prm.grad.clone() for prm in b.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])       # same seed, same masks
    assert all(torch.equal(u, v) for u, v in zip(outs[0][2], outs[1][2]))
    assert not torch.equal(outs[0][0], outs[2][0])                                           # another seed, other masks
    assert all(torch.isfinite(g_).all() for g_ in outs[0][2]) and all(g_.abs().sum() > 0 for g_ in outs[0][2])
    a.train()
    with torch.no_grad():
        ya = a(x, (16, 16))
    assert (outs[0][0] - ya).abs().max().item() > 1e-3          # and the masks do act


def test_block_dropout_matches_the_op_chain_given_the_masks():
    """drop > 0 (Mlp twice, proj) + attn_drop > 0 in one block against the reference's sequence written with torch ops and the
    SAME masks: torch's elementwise masks are reproduced by re-seeding, the attention mask is exported."""
    from rdst_amd import ops
    p_drop, p_attn = 0.2, 0.3
    blk = _block(p_drop, p_attn).train()
    x = _rand((2, 256, 60), 43)
    torch.manual_seed(99)
    xi = x.to(DEV).requires_grad_(True)
    y = blk(xi, (16, 16))
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    # replay the generator: the block draws (in order) the attention seed, the proj mask, the two Mlp masks
    torch.manual_seed(99)
    seed = ops.draw_seed(torch.device(DEV))
    ones = torch.ones(2, 256, 60, device=DEV)
    m_proj = F.dropout(ones, p_drop, True).cpu()
    m_h = F.dropout(torch.ones(2, 256, 120, device=DEV), p_drop, True).cpu()
    m_y = F.dropout(ones, p_drop, True).cpu()
    M = _export_mask(2, 16, 16, 6, 8, p_attn, seed)
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    h = F.layer_norm(xr, (60,), sd["norm1.weight"], sd["norm1.bias"])
    qkv = F.linear(h, sd["attn.qkv.weight"], sd["attn.qkv.bias"]).view(2, 16, 16, 180)
    a = O.window_attention_core(qkv, sd["attn.relative_position_bias_table"], 6, 8, 4, 10 ** -0.5, drop_mult=M).view(2, 256, 60)
    x1 = xr + F.linear(a, sd["attn.proj.weight"], sd["attn.proj.bias"]) * m_proj
    h2 = F.gelu(F.linear(F.layer_norm(x1, (60,), sd["norm2.weight"], sd["norm2.bias"]), sd["mlp.fc1.weight"], sd["mlp.fc1.bias"])) * m_h
    yr = x1 + F.linear(h2, sd["mlp.fc2.weight"], sd["mlp.fc2.bias"]) * m_y
    yr.backward(torch.ones_like(yr))
    assert (y.detach().cpu() - yr.detach()).norm().item() <= 1e-5 * yr.detach().norm().item()
    assert (xi.grad.cpu() - xr.grad).norm().item() <= 1e-5 * xr.grad.norm().item()


def test_network_with_dropout_trains_and_graph_replays_draw_new_masks():
    """make_RDSTSR-level: swin_drop_rate / swin_attn_drop_rate > 0 (rdst_variations.py:1374-1411) through the trainer step, eager
    and replayed from a HIP graph: every replay draws new masks (the seed lives on the device), losses stay finite."""
    from util import build_net
    from rdst_amd.trainer import DPTrainStep
    net = build_net(O.CFG_TINY, drop_rate=0.1, attn_drop=0.1)
    net.load_state_dict(O.make_weights(O.CFG_TINY, 1), strict=True)
    net.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 32, 32, generator=g).to(DEV)
    tgt = torch.rand(2, 1, 128, 128, generator=g).to(DEV)
    tr = DPTrainStep(net, lr=0.0, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, graph=True, graph_warmup=1)
    losses = [float(tr.step(x, tgt)) for _ in range(6)]
    assert tr.graph is not None
    assert all(np.isfinite(l) for l in losses)
    assert len({round(l, 7) for l in losses[2:]}) > 1, losses           # lr = 0: only the masks change from replay to replay
    net.eval()
    with torch.no_grad():
        y1, y2 = net(x), net(x)
    assert torch.equal(y1, y2)
