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
    (64, 126, 254),             # the largest supported widths
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
    tol = 2e-2   # bf16 operands, fp32 accumulation; the same bound test_ops_gpu.py uses for the unfused bf16 kernels
    assert _rel(dx, gref[0]) <= tol
    for got, want, name in zip(G, (gref[3], gref[4], gref[5], gref[6], gref[1], gref[2]),
                               ("dW1", "db1", "dW2", "db2", "dln_w", "dln_b")):
        assert torch.isfinite(got).all(), name
        assert _rel(got, want) <= tol, name


def test_mlp_unsupported_shapes_say_so():
    from rdst_amd import _lib
    lib = _lib.load()
    assert lib.rdst_mlp_fused_supported(60, 120, _lib.F32) == 0     # fp32 parity mode composes K3 kernels
    assert lib.rdst_mlp_fused_supported(128, 256, _lib.BF16) == 0   # no room for the ones column
    assert lib.rdst_mlp_fused_supported(60, 200, _lib.BF16) == 0    # hidden width beyond the wave count
    assert lib.rdst_mlp_fused_supported(61, 120, _lib.BF16) == 0    # odd rows are not dword aligned
