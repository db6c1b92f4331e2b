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


@pytest.mark.parametrize("K", [60, 90, 120])
@pytest.mark.parametrize("M", [32 * 37 + 5, 40013])
def test_lnlin3_bwd_second_addend(K, M):
    """rdst_ln_linear_bwd2: dX = dX_add + dX_add2 + LN'(dY W) with dX_add2 a STRIDED slice (the first K channels of a
    (M, K + 30) gradient buffer: what a dense join hands back for its prefix, rdst_variations.py:339-340) — equal to the
    kernel without it plus a plain add, and to torch autograd; a shape the one-pass kernel does not cover answers
    RDST_ENOTSUP with nothing written."""
    from rdst_amd import _lib
    lib = _lib.load()
    N = 3 * K
    x = rand((M, K), 1).bfloat16().float()
    gy = rand((M, N), 2).bfloat16().float()
    add = rand((M, K), 3).bfloat16().float()
    wide = rand((M, K + 30), 8).bfloat16().float()
    w, b = rand((N, K), 4, K ** -0.5), 0.1 * rand((N,), 5)
    lw, lb = 1 + 0.1 * rand((K,), 6), 0.1 * rand((K,), 7)
    xr, wr, br, lwr, lbr = [t.clone().requires_grad_(True) for t in (x, w, b, lw, lb)]
    F.linear(F.layer_norm(xr, (K,), lwr, lbr, 1e-5), wr, br).backward(gy)
    want_dx = xr.grad + add + wide[:, :K]

    xg, gyg, addg, wideg = (t.to(DEV).bfloat16() for t in (x, gy, add, wide))
    xf = xg.float()
    stats = torch.stack([xf.mean(-1), (xf.var(-1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
    P = [t.to(DEV).contiguous() for t in (w, lw, lb)]
    nws = lib.rdst_ln_linear_bwd_workspace(M, K, N)
    wsp = torch.empty(nws, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def run(fn, extra, dx):
        dW, db = torch.empty_like(P[0]), torch.empty(N, device=DEV)
        dlw, dlb = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
        rc = fn(xg.data_ptr(), K, P[1].data_ptr(), P[2].data_ptr(), stats.data_ptr(), 0, P[0].data_ptr(), gyg.data_ptr(), N,
                dx.data_ptr(), K, addg.data_ptr(), K, dW.data_ptr(), db.data_ptr(), dlw.data_ptr(), dlb.data_ptr(),
                wsp.data_ptr(), nws, M, K, N, 1.0, _lib.BF16, st, *extra)
        torch.cuda.synchronize()
        return rc, dW, db, dlw, dlb

    dx2 = torch.full_like(xg, float("nan"))
    rc, dW2, db2, dlw2, dlb2 = run(lib.rdst_ln_linear_bwd2, (wideg.data_ptr(), K + 30), dx2)
    _lib.check(rc, "rdst_ln_linear_bwd2")
    dx1 = torch.full_like(xg, float("nan"))
    rc, dW1, db1, dlw1, dlb1 = run(lib.rdst_ln_linear_bwd, (), dx1)
    _lib.check(rc, "rdst_ln_linear_bwd")
    assert torch.isfinite(dx2).all()
    assert _rel(dx2, want_dx) <= 8e-3
    # against the kernel without the second addend + the add autograd would have launched: two bf16 roundings either way
    ref = (dx1.float() + wideg[:, :K].float()).bfloat16()
    assert (dx2.float() - ref.float()).abs().max().item() <= 2 ** -7 * max(1.0, ref.float().abs().max().item())
    assert torch.equal(dW1, dW2) and torch.equal(db1, db2) and torch.equal(dlw1, dlw2) and torch.equal(dlb1, dlb2)
    # a shape outside the one-pass kernel's set: refused, dX untouched
    Kb, Nb = 64, 64
    xb, gb = torch.zeros(M, Kb, device=DEV, dtype=torch.bfloat16), torch.zeros(M, Nb, device=DEV, dtype=torch.bfloat16)
    sb = torch.zeros(M, 2, device=DEV)
    wb, ob = torch.zeros(Nb, Kb, device=DEV), torch.ones(Kb, device=DEV)
    dxb = torch.full((M, Kb), 7.0, device=DEV, dtype=torch.bfloat16)
    nb = lib.rdst_ln_linear_bwd_workspace(M, Kb, Nb)
    wspb = torch.empty(nb, dtype=torch.uint8, device=DEV)
    g = [torch.empty(Nb, Kb, device=DEV), torch.empty(Nb, device=DEV), torch.empty(Kb, device=DEV), torch.empty(Kb, device=DEV)]
    rc = lib.rdst_ln_linear_bwd2(xb.data_ptr(), Kb, ob.data_ptr(), ob.data_ptr(), sb.data_ptr(), 0, wb.data_ptr(), gb.data_ptr(),
                                 Nb, dxb.data_ptr(), Kb, None, 0, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                 g[3].data_ptr(), wspb.data_ptr(), nb, M, Kb, Nb, 1.0, _lib.BF16, st, xb.data_ptr(), Kb)
    torch.cuda.synchronize()
    assert rc == _lib.ENOTSUP and (dxb == 7.0).all()
