"""K1/K2 parity: the HIP window-attention kernels (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import rdst_oracle as O

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))


CASES = [
    # B, H, W, C, heads, ws, shift
    (2, 16, 16, 60, 6, 8, 0),
    (2, 16, 16, 60, 6, 8, 4),
    (1, 16, 24, 90, 6, 8, 4),     # non-square, head dim 15
    (1, 24, 16, 120, 6, 8, 4),    # head dim 20
    (1, 32, 32, 60, 6, 16, 8),    # window 16 (N = 256)
    (1, 32, 48, 48, 6, 16, 0),
    (3, 8, 8, 48, 6, 8, 0),       # one window per image
    (2, 8, 8, 60, 6, 8, 4),       # shifted, single window: every region in one window
    (2, 8, 12, 72, 6, 4, 2),      # window 4 (RDSTSR ctor default), head dim 12
]


@pytest.mark.parametrize("B,H,W,C,heads,ws,shift", CASES)
def test_wattn_fwd_bwd_fp32(B, H, W, C, heads, ws, shift):
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    scale = (C // heads) ** -0.5
    qkv = _rand((B, H, W, 3 * C), 1)
    table = _rand(((2 * ws - 1) ** 2, heads), 2, 0.5)
    gout = _rand((B, H, W, C), 3)

    q_ref = qkv.clone().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout)

    q = qkv.to(dev).requires_grad_(True)
    t = table.to(dev).requires_grad_(True)
    o = ops.window_attention(q, t, H, W, heads, ws, shift, scale)
    o.backward(gout.to(dev))
    torch.cuda.synchronize()

    # fp32 tolerance: |d| <= 2e-5 abs on O(1) outputs (different summation order, expf vs Sleef exp)
    assert (o.cpu() - o_ref).abs().max().item() <= 2e-5
    assert (q.grad.cpu() - q_ref.grad).abs().max().item() <= 5e-5
    rel = (t.grad.cpu() - t_ref.grad).norm().item() / t_ref.grad.norm().item()
    assert rel <= 1e-5, rel


@pytest.mark.parametrize("B,H,W,C,heads,ws,shift", CASES[:5])
def test_wattn_fwd_bwd_bf16(B, H, W, C, heads, ws, shift):
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    scale = (C // heads) ** -0.5
    qkv = _rand((B, H, W, 3 * C), 1).bfloat16()
    table = _rand(((2 * ws - 1) ** 2, heads), 2, 0.5)
    gout = _rand((B, H, W, C), 3).bfloat16()

    q_ref = qkv.float().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout.float())

    q = qkv.to(dev).requires_grad_(True)
    t = table.to(dev).requires_grad_(True)
    o = ops.window_attention(q, t, H, W, heads, ws, shift, scale)
    o.backward(gout.to(dev))
    torch.cuda.synchronize()
    # bf16 I/O (8 mantissa bits): outputs are rounded once, inputs identical
    assert (o.float().cpu() - o_ref).abs().max().item() <= 2e-2
    assert (q.grad.float().cpu() - q_ref.grad).abs().max().item() <= 6e-2
    rel = (t.grad.cpu() - t_ref.grad).norm().item() / t_ref.grad.norm().item()
    assert rel <= 2e-2, rel


def test_wattn_strided_rows():
    """qkv living inside a wider buffer (ld > 3C) gives the same result."""
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    B, H, W, C, heads, ws, shift = 1, 16, 16, 60, 6, 8, 4
    scale = 10 ** -0.5
    big = _rand((B, H, W, 3 * C + 20), 5).to(dev)
    table = _rand((225, heads), 6, 0.5).to(dev)
    a = ops.window_attention(big[..., 4:4 + 3 * C], table, H, W, heads, ws, shift, scale)
    b = ops.window_attention(big[..., 4:4 + 3 * C].contiguous(), table, H, W, heads, ws, shift, scale)
    assert torch.equal(a, b)


def test_wattn_bad_args():
    from rdst_amd import ops, _lib
    dev = torch.device("cuda:0")
    qkv = torch.zeros(1, 12, 16, 180, device=dev)
    table = torch.zeros(225, 6, device=dev)
    with pytest.raises(_lib.HipError):
        ops.window_attention(qkv, table, 12, 16, 6, 8, 0, 1.0)   # H not a multiple of ws
    with pytest.raises(RuntimeError):
        ops.window_attention(qkv.cpu(), table.cpu(), 16, 16, 6, 8, 0, 1.0)  # CPU tensors: no fallback


WS16_CASES = [
    # B, H, W, C, shift
    (1, 32, 32, 60, 0),
    (1, 32, 32, 60, 8),      # 2 x 2 windows, every mask case (last row, last column, corner)
    (1, 32, 48, 90, 8),      # head dim 15 (12-byte chunks), non-square
    (2, 48, 32, 120, 8),     # head dim 20
    (1, 16, 16, 120, 5),     # one window per image: all four regions in it, odd shift
    (2, 64, 64, 60, 8),      # 32 windows: the XCD-aware block mapping (window count % 8 == 0)
]


@pytest.mark.parametrize("B,H,W,C,shift", WS16_CASES)
def test_wattn16_bf16(B, H, W, C, shift):
    """16 x 16 windows on the matrix-core kernels (wattn16_mfma.hip) vs the oracle, forward and backward."""
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    heads, ws = 6, 16
    scale = (C // heads) ** -0.5
    qkv = _rand((B, H, W, 3 * C), 11).bfloat16()
    table = _rand(((2 * ws - 1) ** 2, heads), 12, 0.5)
    gout = _rand((B, H, W, C), 13).bfloat16()

    q_ref = qkv.float().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout.float())

    q = qkv.to(dev).requires_grad_(True)
    t = table.to(dev).requires_grad_(True)
    o = ops.window_attention(q, t, H, W, heads, ws, shift, scale)
    o.backward(gout.to(dev))
    torch.cuda.synchronize()
    assert (o.float().cpu() - o_ref).abs().max().item() <= 2e-2
    assert (o.float().cpu() - o_ref).norm().item() <= 5e-3 * o_ref.norm().item()
    assert (q.grad.float().cpu() - q_ref.grad).abs().max().item() <= 6e-2
    assert (q.grad.float().cpu() - q_ref.grad).norm().item() <= 2e-2 * q_ref.grad.norm().item()
    rel = (t.grad.cpu() - t_ref.grad).norm().item() / t_ref.grad.norm().item()
    assert rel <= 2e-2, rel


@pytest.mark.parametrize("B,H,W,C,shift", WS16_CASES)
def test_wattn16_fp32_mfma_vs_oracle(B, H, W, C, shift):
    """16 x 16 windows in the reference's own arithmetic (exact-fp32 matrix-core kernels, wattn16_f32.hip: BASELINE
    configs[3] without bf16) vs the oracle, forward and backward: relative L2 <= 2e-6 on the output and on every gradient
    (fp32 sums in a different order, expf vs the CPU's exp)."""
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    heads, ws = 6, 16
    scale = (C // heads) ** -0.5
    qkv = _rand((B, H, W, 3 * C), 21)
    table = _rand(((2 * ws - 1) ** 2, heads), 22, 0.5)
    gout = _rand((B, H, W, C), 23)

    q_ref = qkv.clone().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout)

    q = qkv.to(dev).requires_grad_(True)
    t = table.to(dev).requires_grad_(True)
    o = ops.window_attention(q, t, H, W, heads, ws, shift, scale)
    o.backward(gout.to(dev))
    torch.cuda.synchronize()

    def rel(a, b):
        return (a.cpu().double() - b.double()).norm().item() / b.double().norm().item()

    assert rel(o, o_ref) <= 2e-6, rel(o, o_ref)
    assert rel(q.grad, q_ref.grad) <= 2e-6, rel(q.grad, q_ref.grad)
    assert rel(t.grad, t_ref.grad) <= 2e-6, rel(t.grad, t_ref.grad)
    assert (o.cpu() - o_ref).abs().max().item() <= 2e-5
    assert (q.grad.cpu() - q_ref.grad).abs().max().item() <= 5e-5
    # the three sections separately (a wrong dK / dV of small norm must not hide behind dQ)
    for s3, name in enumerate(("dq", "dk", "dv")):
        a, b_ = q.grad[..., s3 * C:(s3 + 1) * C], q_ref.grad[..., s3 * C:(s3 + 1) * C]
        assert rel(a, b_) <= 3e-6, (name, rel(a, b_))


def test_wattn16_fp32_is_bit_deterministic():
    """The exact-fp32 window-16 kernels sum d(table) in a fixed order (lane-shift diagonal sums, then one thread per entry, then
    the batched slab sum): two runs give the same bits for the output and every gradient."""
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    B, H, W, C, shift = 2, 64, 64, 90, 8
    qkv = _rand((B, H, W, 3 * C), 31).to(dev)
    table = _rand((961, 6), 32, 0.5).to(dev)
    gout = _rand((B, H, W, C), 33).to(dev)
    res = []
    for _ in range(2):
        q = qkv.clone().requires_grad_(True)
        t = table.clone().requires_grad_(True)
        o = ops.window_attention(q, t, H, W, 6, 16, shift, 15 ** -0.5)
        o.backward(gout)
        torch.cuda.synchronize()
        res.append((o.detach().clone(), q.grad.clone(), t.grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------------------------
# K2, the (window, 6 heads) kernel with the LDS-DMA ring (csrc/wattn_bwd_pair.hip, default for C = 90 / 120): the cases the
# ring's prefetch guards and the loader / storer split are written for
# ------------------------------------------------------------------------------------------------------------------
def _k2_case(B, H, W, C, shift, seed):
    heads, ws = 6, 8
    scale = (C // heads) ** -0.5
    qkv = _rand((B, H, W, 3 * C), seed).bfloat16()
    table = _rand(((2 * ws - 1) ** 2, heads), seed + 1, 0.5)
    gout = _rand((B, H, W, C), seed + 2).bfloat16()
    return heads, ws, scale, qkv, table, gout


@pytest.mark.parametrize("C", [90, 120])
@pytest.mark.parametrize("B,H,W,shift", [(257, 8, 8, 0), (3, 80, 80, 1), (1, 136, 24, 3), (7, 64, 72, 7), (2, 8, 16, 5)])
def test_k2_pair_ragged_groups_and_shifts_vs_oracle(C, B, H, W, shift):
    """Window counts that are not multiples of the 256 workgroups (257: one group with 2 items, 255 with 1; 300; 51; 504;
    4: fewer windows than ring slots), so the NBUF - 1 items a group keeps in flight run past its last window, and shifts
    other than 0 / 4 (every split of the wrapped window rows)."""
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    heads, ws, scale, qkv, table, gout = _k2_case(B, H, W, C, shift, 31 + C + shift)
    q_ref = qkv.float().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale).backward(gout.float())
    q = qkv.to(dev).requires_grad_(True)
    t = table.to(dev).requires_grad_(True)
    ops.window_attention(q, t, H, W, heads, ws, shift, scale).backward(gout.to(dev))
    torch.cuda.synchronize()
    assert torch.isfinite(q.grad.float()).all()
    assert (q.grad.float().cpu() - q_ref.grad).abs().max().item() <= 6e-2
    rel_q = (q.grad.float().cpu() - q_ref.grad).norm().item() / q_ref.grad.norm().item()
    rel_t = (t.grad.cpu() - t_ref.grad).norm().item() / t_ref.grad.norm().item()
    assert rel_q <= 1e-2 and rel_t <= 2e-2, (rel_q, rel_t)


@pytest.mark.parametrize("C", [90, 120])
def test_k2_pair_equals_the_strided_row_fallback(C):
    """Rows that are not dense (qkv inside a wider buffer) make the pair kernel refuse (RDST_ENOTSUP) and the (window, 12 wave)
    kernel of rounds 1-3 run: same function, two kernels — dqkv within bf16 rounding of each other (both round each gradient
    once), d(table) within the bf16 bound of the other kernel tests (the older kernel sums dS after its bf16 round trip through
    LDS, the pair kernel sums the fp32 accumulators: measured 2.4e-3), and each bit-deterministic run to run."""
    from rdst_amd import ops
    dev = torch.device("cuda:0")
    B, H, W, shift = 5, 64, 72, 4
    heads, ws, scale, qkv, table, gout = _k2_case(B, H, W, C, shift, 77 + C)
    big = torch.zeros(B, H, W, 3 * C + 8, dtype=torch.bfloat16)
    big[..., :3 * C] = qkv
    res = []
    for src in (qkv.to(dev), big.to(dev)[..., :3 * C]):
        for _ in range(2):
            q = src.detach().requires_grad_(True)
            t = table.to(dev).requires_grad_(True)
            ops.window_attention(q, t, H, W, heads, ws, shift, scale).backward(gout.to(dev))
            torch.cuda.synchronize()
            res.append((q.grad.float().cpu()[..., :3 * C].clone(), t.grad.cpu().clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])       # dense rows (pair kernel): run to run
    assert torch.equal(res[2][0], res[3][0]) and torch.equal(res[2][1], res[3][1])       # strided rows (fallback): run to run
    dq = (res[0][0] - res[2][0]).abs().max().item()
    assert dq <= 2 ** -7 * max(1.0, res[0][0].abs().max().item()), dq                       # one bf16 ulp of the largest gradient
    assert (res[0][1] - res[2][1]).norm().item() <= 8e-3 * res[0][1].norm().item()


# ------------------------------------------------------------------------------------------------------------------
# Window 16 with the forward's row statistics kept (rdst_wattn_fwd_lse / rdst_wattn_bwd_lse, round 5): the backward's first
# pass streams its key tiles; delta = rowsum(dO o O) from the forward's output
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [60, 90, 120])
@pytest.mark.parametrize("B,H,W,shift", [(1, 16, 16, 0), (2, 32, 48, 8), (1, 48, 32, 5), (3, 32, 32, 0)])
def test_wattn16_lse_path_vs_oracle_and_plain_path(C, B, H, W, shift):
    from rdst_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    heads, ws = 6, 16
    scale = (C // heads) ** -0.5
    M = B * H * W
    qkv = _rand((B, H, W, 3 * C), 5 + C + shift).bfloat16()
    table = _rand(((2 * ws - 1) ** 2, heads), 6 + C, 0.5)
    gout = _rand((B, H, W, C), 7 + C).bfloat16()
    q_ref = qkv.float().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout.float())

    qg, tg, gg = qkv.to(dev), table.to(dev), gout.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    out = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=dev)
    out0 = torch.empty_like(out)
    nlse = torch.full((M, heads), float("nan"), device=dev)
    _lib.check(lib.rdst_wattn_fwd_lse(qg.data_ptr(), 3 * C, tg.data_ptr(), out.data_ptr(), C, nlse.data_ptr(), B, H, W, C, heads, ws,
                                      shift, scale, _lib.BF16, st), "fwd_lse")
    _lib.check(lib.rdst_wattn_fwd(qg.data_ptr(), 3 * C, tg.data_ptr(), None, 0, out0.data_ptr(), C, B, H, W, C, heads, ws, shift, scale,
                                  _lib.BF16, st), "fwd")
    torch.cuda.synchronize()
    assert torch.equal(out, out0)                                   # the same forward, plus the statistics
    assert torch.isfinite(nlse).all()
    assert (out.float().cpu().reshape(o_ref.shape) - o_ref.detach()).abs().max().item() <= 2e-2

    nb = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
    res = []
    for which in ("lse", "lse", "plain"):
        dq = torch.full((M, 3 * C), float("nan"), dtype=torch.bfloat16, device=dev)
        dt = torch.full_like(tg, float("nan"))
        wsp = torch.empty(nb, dtype=torch.uint8, device=dev)
        if which == "lse":
            _lib.check(lib.rdst_wattn_bwd_lse(qg.data_ptr(), 3 * C, tg.data_ptr(), gg.data_ptr(), C, out.data_ptr(), C, nlse.data_ptr(),
                                              dq.data_ptr(), 3 * C, dt.data_ptr(), wsp.data_ptr(), nb, B, H, W, C, heads, ws, shift,
                                              scale, _lib.BF16, st), "bwd_lse")
        else:
            _lib.check(lib.rdst_wattn_bwd(qg.data_ptr(), 3 * C, tg.data_ptr(), None, 0, gg.data_ptr(), C, dq.data_ptr(), 3 * C,
                                          dt.data_ptr(), wsp.data_ptr(), nb, B, H, W, C, heads, ws, shift, scale, _lib.BF16, st), "bwd")
        torch.cuda.synchronize()
        res.append((dq.float().cpu().reshape(q_ref.shape), dt.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])        # bit-deterministic run to run
    for name, (dq, dt) in (("lse", res[0]), ("plain", res[2])):
        assert torch.isfinite(dq).all() and torch.isfinite(dt).all(), name
        rel_q = (dq - q_ref.grad).norm().item() / q_ref.grad.norm().item()
        rel_t = (dt - t_ref.grad).norm().item() / t_ref.grad.norm().item()
        assert (dq - q_ref.grad).abs().max().item() <= 6e-2 and rel_q <= 1.2e-2 and rel_t <= 2e-2, (name, rel_q, rel_t)


def test_wattn_lse_entry_points_refuse_other_shapes():
    from rdst_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    C, heads = 60, 6
    qkv = torch.zeros(1, 16, 16, 3 * C, dtype=torch.bfloat16, device=dev)
    table = torch.zeros(225, heads, device=dev)
    out = torch.empty(256, C, dtype=torch.bfloat16, device=dev)
    nlse = torch.empty(256, heads, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.rdst_wattn_fwd_lse(qkv.data_ptr(), 3 * C, table.data_ptr(), out.data_ptr(), C, nlse.data_ptr(), 1, 16, 16, C, heads, 8, 0,
                                0.3, _lib.BF16, st)
    assert rc == _lib.ENOTSUP            # window 8 is served by other kernels: the caller uses rdst_wattn_fwd
