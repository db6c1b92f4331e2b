"""K7 parity: the fused Mlp kernels (through the C ABI) vs plain torch fp32 autograd of
y = x + fc2(GELU(fc1(LayerNorm(x))))  (swin_transformer_sr.py:23-29, :272)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O
from util import rand

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [  # M, C, hid
    (32 * 37 + 5, 60, 120),     # ragged last tile
    (4096, 90, 180),
    (2048 + 32, 120, 240),
    (300, 62, 100),             # hid != 2C, odd pack tails
    (64, 124, 232),             # about the largest widths whose two weight images fit the 160 KB of LDS
]


def _reference(x, lw, lb, w1, b1, w2, b2, gy):
    leaves = [t.clone().requires_grad_(True) for t in (x, lw, lb, w1, b1, w2, b2)]
    xr, lwr, lbr, w1r, b1r, w2r, b2r = leaves
    h = F.linear(F.layer_norm(xr, (x.shape[-1],), lwr, lbr, 1e-5), w1r, b1r)
    y = xr + F.linear(O.gelu(h), w2r, b2r)
    y.backward(gy)
    return y.detach(), [t.grad for t in leaves]


def _inputs(M, C, hid):
    x = rand((M, C), 1).bfloat16().float()
    gy = rand((M, C), 2).bfloat16().float()
    lw, lb = 1 + 0.1 * rand((C,), 3), 0.1 * rand((C,), 4)
    w1, b1 = rand((hid, C), 5, C ** -0.5), 0.1 * rand((hid,), 6)
    w2, b2 = rand((C, hid), 7, hid ** -0.5), 0.1 * rand((C,), 8)
    return x, gy, lw, lb, w1, b1, w2, b2


def _rel(a, b):
    return (a.float().cpu() - b).norm().item() / max(b.norm().item(), 1e-12)


@pytest.mark.parametrize("M,C,hid", CASES)
def test_mlp_bwd_fused(M, C, hid):
    from rdst_amd import _lib
    lib = _lib.load()
    assert lib.rdst_mlp_fused_supported(C, hid, _lib.BF16) == 1
    x, gy, lw, lb, w1, b1, w2, b2 = _inputs(M, C, hid)
    _, gref = _reference(x, lw, lb, w1, b1, w2, b2, gy)

    xg, gyg = x.to(DEV).bfloat16(), gy.to(DEV).bfloat16()
    P = [t.to(DEV).contiguous() for t in (lw, lb, w1, b1, w2)]
    xf = xg.float()
    mean = xf.mean(-1)
    rstd = (xf.var(-1, unbiased=False) + 1e-5).rsqrt()
    stats = torch.stack([mean, rstd], dim=1).contiguous()
    dx = torch.empty_like(xg)
    G = [torch.full_like(t, float("nan")) for t in (P[2], P[3], P[4], b2.to(DEV), P[0], P[1])]   # dW1 db1 dW2 db2 dlw dlb
    nb = lib.rdst_mlp_bwd_workspace(M, C, hid)
    wsp = torch.empty(nb, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.rdst_mlp_bwd(xg.data_ptr(), C, P[0].data_ptr(), P[1].data_ptr(), stats.data_ptr(), P[2].data_ptr(),
                                P[3].data_ptr(), P[4].data_ptr(), gyg.data_ptr(), C, dx.data_ptr(), C, G[0].data_ptr(),
                                G[1].data_ptr(), G[2].data_ptr(), G[3].data_ptr(), G[4].data_ptr(), G[5].data_ptr(),
                                wsp.data_ptr(), nb, M, C, hid, _lib.BF16, st), "rdst_mlp_bwd")
    torch.cuda.synchronize()
    tol = 8e-3   # bf16 operands, fp32 accumulation; the same bound test_ops_gpu.py uses for the unfused bf16 kernels
    assert _rel(dx, gref[0]) <= tol
    for got, want, name in zip(G, (gref[3], gref[4], gref[5], gref[6], gref[1], gref[2]),
                               ("dW1", "db1", "dW2", "db2", "dln_w", "dln_b")):
        assert torch.isfinite(got).all(), name
        assert _rel(got, want) <= tol, name


@pytest.mark.parametrize("M,C,hid", CASES)
def test_mlp_fwd_fused(M, C, hid):
    from rdst_amd import _lib
    lib = _lib.load()
    x, gy, lw, lb, w1, b1, w2, b2 = _inputs(M, C, hid)
    yref, _ = _reference(x, lw, lb, w1, b1, w2, b2, gy)
    xg = x.to(DEV).bfloat16()
    P = [t.to(DEV).contiguous() for t in (lw, lb, w1, b1, w2, b2)]
    y = torch.full_like(xg, float("nan"))
    stats = torch.full((M, 2), float("nan"), dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    nwf = lib.rdst_mlp_fwd_workspace(C, hid)
    wsf = torch.empty(max(nwf, 16), dtype=torch.uint8, device=DEV)
    _lib.check(lib.rdst_mlp_fwd(xg.data_ptr(), C, P[0].data_ptr(), P[1].data_ptr(), P[2].data_ptr(), P[3].data_ptr(),
                                P[4].data_ptr(), P[5].data_ptr(), y.data_ptr(), C, stats.data_ptr(), wsf.data_ptr(), nwf, M, C, hid, _lib.BF16,
                                st), "rdst_mlp_fwd")
    torch.cuda.synchronize()
    # bf16 operands, fp32 accumulation, bf16 output: the bound test_ops_gpu.py uses for the unfused bf16 kernels
    assert (y.float().cpu() - yref).abs().max().item() <= 3e-2 * max(1.0, yref.abs().max().item())
    assert _rel(y, yref) <= 1e-2
    xf = x.double()
    mean, var = xf.mean(-1), xf.var(-1, unbiased=False)
    assert (stats[:, 0].double().cpu() - mean).abs().max().item() <= 1e-5
    assert ((stats[:, 1].double().cpu() - (var + 1e-5).rsqrt()) / (var + 1e-5).rsqrt()).abs().max().item() <= 1e-5


def test_swin_block_fused_vs_composed():
    """The block with K7 on and off (same process, module switch) agrees to bf16 noise, forward and backward."""
    from rdst_amd import ops
    C, hid, H, W, heads, ws = 60, 120, 16, 16, 6, 8
    torch.manual_seed(0)
    x = torch.randn(2, H * W, C, device=DEV).bfloat16()
    prm = [1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C), torch.randn(3 * C, C) * C ** -0.5, 0.1 * torch.randn(3 * C),
           0.02 * torch.randn((2 * ws - 1) ** 2, heads), torch.randn(C, C) * C ** -0.5, 0.1 * torch.randn(C),
           1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C), torch.randn(hid, C) * C ** -0.5, 0.1 * torch.randn(hid),
           torch.randn(C, hid) * hid ** -0.5, 0.1 * torch.randn(C)]
    gy = torch.randn(2, H * W, C, device=DEV).bfloat16()
    res = []
    keep = ops.MLP_FUSED
    try:
        for fused in (True, False):
            ops.MLP_FUSED = fused
            xs = x.clone().requires_grad_(True)
            ps = [t.to(DEV).requires_grad_(True) for t in prm]
            y = ops.swin_block(xs, *ps, H, W, heads, ws, 4, (C // heads) ** -0.5)
            y.backward(gy)
            torch.cuda.synchronize()
            res.append([y.detach().float(), xs.grad.float()] + [t.grad.float() for t in ps])
    finally:
        ops.MLP_FUSED = keep
    for a, b in zip(*res):
        assert (a - b).norm().item() <= 2e-2 * max(b.norm().item(), 1e-6)


def test_mlp_unsupported_shapes_say_so():
    from rdst_amd import _lib
    lib = _lib.load()
    assert lib.rdst_mlp_fused_supported(60, 120, _lib.F32) == 0     # fp32 parity mode composes K3 kernels
    assert lib.rdst_mlp_fused_supported(128, 256, _lib.BF16) == 0   # no room for the ones column
    assert lib.rdst_mlp_fused_supported(126, 254, _lib.BF16) == 0   # the forward's weight images exceed the LDS
    assert lib.rdst_mlp_fused_supported(60, 200, _lib.BF16) == 0    # hidden width beyond the wave count
    assert lib.rdst_mlp_fused_supported(61, 120, _lib.BF16) == 0    # odd rows are not dword aligned


def test_swin_block_backward_with_misaligned_gradient():
    """the fused Mlp backward refuses a dy whose rows are not dword aligned (a strided gradient slice with an
    odd channel offset); with the forward fused there is no composed fallback, so the block retries on an aligned copy."""
    from rdst_amd import ops
    C, hid, H, W, heads, ws = 60, 120, 16, 16, 6, 8
    torch.manual_seed(1)
    x = torch.randn(2, H * W, C, device=DEV).bfloat16()
    prm = [1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C), torch.randn(3 * C, C) * C ** -0.5, 0.1 * torch.randn(3 * C),
           0.02 * torch.randn((2 * ws - 1) ** 2, heads), torch.randn(C, C) * C ** -0.5, 0.1 * torch.randn(C),
           1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C), torch.randn(hid, C) * C ** -0.5, 0.1 * torch.randn(hid),
           torch.randn(C, hid) * hid ** -0.5, 0.1 * torch.randn(C)]
    from rdst_amd import _lib
    lib = _lib.load()
    real, codes = lib.rdst_mlp_bwd, []

    def spy(*args):
        codes.append(real(*args))
        return codes[-1]

    wide = torch.randn(2, H * W, C + 3, device=DEV).bfloat16()
    res, seen = [], []
    lib.rdst_mlp_bwd = spy
    try:
        for gy in (wide[..., 1:C + 1], wide[..., 1:C + 1].contiguous()):   # rows 2-byte aligned only / compact copy
            del codes[:]
            xs = x.clone().requires_grad_(True)
            ps = [t.to(DEV).requires_grad_(True) for t in prm]
            y = ops.swin_block(xs, *ps, H, W, heads, ws, 0, (C // heads) ** -0.5)
            y.backward(gy)
            torch.cuda.synchronize()
            res.append([xs.grad.float()] + [t.grad.float() for t in ps])
            seen.append(list(codes))
    finally:
        lib.rdst_mlp_bwd = real
    assert seen[0] == [_lib.ENOTSUP, 0], seen   # refused once, accepted on the aligned copy: the retry path really ran
    assert seen[1] == [0], seen
    for a, b in zip(*res):
        assert torch.equal(a, b)
