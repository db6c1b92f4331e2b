"""The re-cut one-pass Linear backward (csrc/lnlin3_mfma.hip) through the C ABI, all nine E1 shapes, vs plain torch fp32
autograd of y = Linear(LayerNorm(x)) (swin_transformer_sr.py:244-268 qkv / proj, rdst_variations.py dense tails):
ragged M (last tile masked), several tiles per workgroup (double-buffered tiles, LayerNorm rows finished one barrier late),
with and without the dX_add operand."""
import pytest
import torch
import torch.nn.functional as F

from util import rand

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(60, 180, True), (90, 270, True), (120, 360, True), (60, 60, False), (90, 90, False), (120, 120, False),
          (60, 30, True), (90, 30, True), (120, 30, True)]


def _rel(a, b):
    return (a.float().cpu() - b).norm().item() / max(b.norm().item(), 1e-12)


@pytest.mark.parametrize("K,N,ln", SHAPES)
@pytest.mark.parametrize("M,with_add", [(32 * 37 + 5, True), (40013, False), (40013, True)])
def test_lnlin3_bwd_vs_torch(K, N, ln, M, with_add):
    from rdst_amd import _lib
    lib = _lib.load()
    x = rand((M, K), 1).bfloat16().float()
    gy = rand((M, N), 2).bfloat16().float()
    add = rand((M, K), 3).bfloat16().float() if with_add else None
    w, b = rand((N, K), 4, K ** -0.5), 0.1 * rand((N,), 5)
    lw, lb = 1 + 0.1 * rand((K,), 6), 0.1 * rand((K,), 7)

    leaves = [t.clone().requires_grad_(True) for t in (x, w, b, lw, lb)]
    xr, wr, br, lwr, lbr = leaves
    h = F.layer_norm(xr, (K,), lwr, lbr, 1e-5) if ln else xr
    F.linear(h, wr, br).backward(gy)
    want_dx = xr.grad + (add if with_add else 0)

    xg, gyg = x.to(DEV).bfloat16(), gy.to(DEV).bfloat16()
    addg = add.to(DEV).bfloat16() if with_add else None
    xf = xg.float()
    stats = torch.stack([xf.mean(-1), (xf.var(-1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
    P = [t.to(DEV).contiguous() for t in (w, lw, lb)]
    dx = torch.full_like(xg, float("nan"))
    dW, db = torch.full_like(P[0], float("nan")), torch.full((N,), float("nan"), device=DEV)
    dlw, dlb = torch.full((K,), float("nan"), device=DEV), torch.full((K,), float("nan"), device=DEV)
    nws = lib.rdst_ln_linear_bwd_workspace(M, K, N)
    wsp = torch.empty(nws, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.rdst_ln_linear_bwd(xg.data_ptr(), K, P[1].data_ptr() if ln else None, P[2].data_ptr() if ln else None,
                                      stats.data_ptr() if ln else None, 0, P[0].data_ptr(), gyg.data_ptr(), N, dx.data_ptr(), K,
                                      addg.data_ptr() if with_add else None, K, dW.data_ptr(), db.data_ptr(),
                                      dlw.data_ptr() if ln else None, dlb.data_ptr() if ln else None, wsp.data_ptr(), nws,
                                      M, K, N, 1.0, _lib.BF16, st), "rdst_ln_linear_bwd")
    torch.cuda.synchronize()
    tol = 8e-3   # bf16 operands, fp32 accumulation, bf16 dX / slabs: the bound of the other bf16 kernel tests
    assert torch.isfinite(dx).all()
    assert _rel(dx, want_dx) <= tol
    assert _rel(dW, wr.grad) <= tol and _rel(db, br.grad) <= tol
    if ln:
        assert _rel(dlw, lwr.grad) <= tol and _rel(dlb, lbr.grad) <= tol
    # every 32-token tile of dX on its own (a wrong buffer / a row finished with the neighbour tile's sums shows here)
    nt = M // 32
    d = (dx.float().cpu() - want_dx)[:nt * 32].view(nt, -1).norm(dim=1) / want_dx[:nt * 32].view(nt, -1).norm(dim=1)
    assert d.max().item() <= 3 * tol
