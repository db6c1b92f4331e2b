"""bf16 parity AT THE SIZES bench.py RUNS (BASELINE.json configs[1]: B 32 of 64x64, 131072 tokens, 2048 windows).

The persistent bf16 kernels keep state across their loops (K2's d(table) accumulators over `win += gridDim.x`, K1's
register-prefetched next window, the fused Mlp / one-pass Linear backward's resident weight-gradient accumulators over
all their 32-token tiles, the conv stripes).  With <= 256 windows / tiles every workgroup runs its loop body once, so
the small cases elsewhere in this suite never exercise that state.  Here every kernel gets 2048 windows / 4096 tiles
(8-16 iterations per workgroup) and is compared with the CPU oracle (window attention: oracle/rdst_oracle.py, which is
pinned to the reference, swin_transformer_sr.py:110-141) or with plain fp32 torch autograd on the CPU of the same op
(Linear / Mlp / conv: nn.Linear, nn.LayerNorm, nn.GELU, nn.Conv2d + PixelShuffle exactly as the reference composes
them, rdst_variations.py:335-341,420-445, common.py:125-136).

Tolerance (bf16 storage = 8 significant bits, fp32 accumulation): relative L2 <= 8e-3 on every output and gradient
(measured values are printed with -s; they sit at 1.4e-3..4e-3: a 2x regression fails), identical bf16-representable
inputs on both sides."""
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O
from util import rand

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 8e-3
B_FULL, HW = 32, 64
M_FULL = B_FULL * HW * HW          # 131072 tokens = 4096 tiles of 32 = 2048 windows of 8x8


def _rel(a, b):
    b = b.float()
    return (a.float().cpu() - b).norm().item() / max(b.norm().item(), 1e-12)


def _bf(t):
    return t.bfloat16().float()


# ------------------------------------------------------------------------------------------------------------------
# K1 / K2 (wattn_mfma_hd.hip, wattn_bwd_mfma_hd.hip): 2048 windows on a 256-workgroup persistent grid
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [60, 90, 120])
@pytest.mark.parametrize("shift", [0, 4])
def test_wattn_bf16_2048_windows_vs_oracle(C, shift):
    from rdst_amd import ops
    heads, ws = 6, 8
    scale = (C // heads) ** -0.5
    qkv = _bf(rand((B_FULL, HW, HW, 3 * C), 100 + C))
    table = rand(((2 * ws - 1) ** 2, heads), 2, 0.5)
    gout = _bf(rand((B_FULL, HW, HW, C), 3))

    q_ref = qkv.clone().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout)

    q = qkv.to(DEV).bfloat16().requires_grad_(True)
    t = table.to(DEV).requires_grad_(True)
    o = ops.window_attention(q, t, HW, HW, heads, ws, shift, scale)
    o.backward(gout.to(DEV).bfloat16())
    torch.cuda.synchronize()
    ro, rq, rt = _rel(o, o_ref.detach()), _rel(q.grad, q_ref.grad), _rel(t.grad, t_ref.grad)
    print(f"\nwattn bf16 C={C} shift={shift}: rel L2 out {ro:.2e}  dqkv {rq:.2e}  dtable {rt:.2e}")
    assert (o.float().cpu() - o_ref.detach()).abs().max().item() <= 3e-2
    assert ro <= TOL and rq <= TOL and rt <= TOL
    # per-window check: no window of the 2048 is off (a wrong prefetch / loop-carried state would hit whole windows)
    d = (o.float().cpu() - o_ref.detach()).view(B_FULL, HW // ws, ws, HW // ws, ws, C)
    per_win = d.pow(2).sum(dim=(2, 4, 5)).sqrt() / o_ref.detach().view(B_FULL, HW // ws, ws, HW // ws, ws, C).pow(2).sum(
        dim=(2, 4, 5)).sqrt()
    assert per_win.max().item() <= TOL
    dq = (q.grad.float().cpu() - q_ref.grad).view(B_FULL, HW // ws, ws, HW // ws, ws, 3 * C)
    per_win = dq.pow(2).sum(dim=(2, 4, 5)).sqrt() / q_ref.grad.view(B_FULL, HW // ws, ws, HW // ws, ws, 3 * C).pow(2).sum(
        dim=(2, 4, 5)).sqrt()
    assert per_win.max().item() <= 2 * TOL


@pytest.mark.parametrize("C,shift", [(60, 4), (90, 0), (120, 4)])
def test_wattn_fp32_2048_windows_vs_oracle(C, shift):
    """The fp32 parity mode's K1 / K2 (wattn_mfma.hip, wattn_bwd_mfma.hip: eight waves, the backward with two (tile, head) units
    on four of them, persistent over 8 windows per workgroup) at the bench size, per-window bounds: a unit or a window that is
    skipped, done twice or read from the wrong rows cannot hide in the norm."""
    from rdst_amd import ops
    heads, ws = 6, 8
    scale = (C // heads) ** -0.5
    qkv = rand((B_FULL, HW, HW, 3 * C), 500 + C)
    table = rand(((2 * ws - 1) ** 2, heads), 2, 0.5)
    gout = rand((B_FULL, HW, HW, C), 3)

    q_ref = qkv.clone().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout)

    q = qkv.to(DEV).requires_grad_(True)
    t = table.to(DEV).requires_grad_(True)
    o = ops.window_attention(q, t, HW, HW, heads, ws, shift, scale)
    o.backward(gout.to(DEV))
    torch.cuda.synchronize()
    ro, rq, rt = _rel(o, o_ref.detach()), _rel(q.grad, q_ref.grad), _rel(t.grad, t_ref.grad)
    print(f"\nwattn fp32 C={C} shift={shift}: rel L2 out {ro:.2e}  dqkv {rq:.2e}  dtable {rt:.2e}")
    assert ro <= 2e-6 and rq <= 2e-6 and rt <= 3e-6     # fp32 arithmetic on both sides (measured 2e-7 .. 3e-7)
    nw = HW // ws
    for got, want, width in ((o, o_ref.detach(), C), (q.grad, q_ref.grad, 3 * C)):
        d = (got.float().cpu() - want).view(B_FULL, nw, ws, nw, ws, width)
        per_win = d.pow(2).sum(dim=(2, 4, 5)).sqrt() / want.view(B_FULL, nw, ws, nw, ws, width).pow(2).sum(dim=(2, 4, 5)).sqrt()
        assert per_win.max().item() <= 5e-6


@pytest.mark.parametrize("C,shift", [(60, 8), (90, 0), (120, 8)])
def test_wattn16_bf16_bench_size_vs_oracle(C, shift):
    """Window 16 at the size of `bench.py --config ws16` (8 x 128 x 128 tokens = 512 windows = 1536 workgroups of
    wattn16_mfma.hip), per-window error bounds so that a misplaced window / head pair cannot hide in the norm."""
    from rdst_amd import ops
    heads, ws, B, HW16 = 6, 16, 8, 128
    scale = (C // heads) ** -0.5
    qkv = _bf(rand((B, HW16, HW16, 3 * C), 300 + C))
    table = rand(((2 * ws - 1) ** 2, heads), 2, 0.5)
    gout = _bf(rand((B, HW16, HW16, C), 3))

    q_ref = qkv.clone().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout)

    q = qkv.to(DEV).bfloat16().requires_grad_(True)
    t = table.to(DEV).requires_grad_(True)
    o = ops.window_attention(q, t, HW16, HW16, heads, ws, shift, scale)
    o.backward(gout.to(DEV).bfloat16())
    torch.cuda.synchronize()
    ro, rq, rt = _rel(o, o_ref.detach()), _rel(q.grad, q_ref.grad), _rel(t.grad, t_ref.grad)
    print(f"\nwattn16 bf16 C={C} shift={shift}: rel L2 out {ro:.2e}  dqkv {rq:.2e}  dtable {rt:.2e}")
    assert ro <= TOL and rq <= TOL and rt <= TOL
    nw = HW16 // ws
    for got, want, width in ((o, o_ref.detach(), C), (q.grad, q_ref.grad, 3 * C)):
        d = (got.float().cpu() - want).view(B, nw, ws, nw, ws, width)
        per_win = d.pow(2).sum(dim=(2, 4, 5)).sqrt() / want.view(B, nw, ws, nw, ws, width).pow(2).sum(dim=(2, 4, 5)).sqrt()
        assert per_win.max().item() <= 2 * TOL
    # run-to-run determinism of the d(table) sums (no atomics in the kernel)
    q2 = qkv.to(DEV).bfloat16().requires_grad_(True)
    t2 = table.to(DEV).requires_grad_(True)
    ops.window_attention(q2, t2, HW16, HW16, heads, ws, shift, scale).backward(gout.to(DEV).bfloat16())
    torch.cuda.synchronize()
    assert torch.equal(t.grad, t2.grad) and torch.equal(q.grad, q2.grad)


@pytest.mark.parametrize("C,shift", [(60, 8), (120, 0)])
def test_wattn16_fp32_bench_size_vs_oracle(C, shift):
    """The exact-fp32 window-16 kernels (wattn16_f32.hip) at the size of `bench.py --config ws16` (8 x 128 x 128 tokens = 512
    windows = 3072 workgroups, the XCD-aware block mapping): rel L2 <= 2e-6 (3e-6 on d(table)), per-window bounds 5e-6."""
    from rdst_amd import ops
    heads, ws, B, HW16 = 6, 16, 8, 128
    scale = (C // heads) ** -0.5
    qkv = rand((B, HW16, HW16, 3 * C), 400 + C)
    table = rand(((2 * ws - 1) ** 2, heads), 2, 0.5)
    gout = rand((B, HW16, HW16, C), 3)
    q_ref = qkv.clone().requires_grad_(True)
    t_ref = table.clone().requires_grad_(True)
    o_ref = O.window_attention_core(q_ref, t_ref, heads, ws, shift, scale)
    o_ref.backward(gout)
    q = qkv.to(DEV).requires_grad_(True)
    t = table.to(DEV).requires_grad_(True)
    o = ops.window_attention(q, t, HW16, HW16, heads, ws, shift, scale)
    o.backward(gout.to(DEV))
    torch.cuda.synchronize()
    ro, rq, rt = _rel(o, o_ref.detach()), _rel(q.grad, q_ref.grad), _rel(t.grad, t_ref.grad)
    print(f"\nwattn16 fp32 C={C} shift={shift}: rel L2 out {ro:.2e}  dqkv {rq:.2e}  dtable {rt:.2e}")
    assert ro <= 2e-6 and rq <= 2e-6 and rt <= 3e-6
    nw = HW16 // ws
    for got, want, width in ((o, o_ref.detach(), C), (q.grad, q_ref.grad, 3 * C)):
        d = (got.float().cpu() - want).view(B, nw, ws, nw, ws, width)
        per_win = d.pow(2).sum(dim=(2, 4, 5)).sqrt() / want.view(B, nw, ws, nw, ws, width).pow(2).sum(dim=(2, 4, 5)).sqrt()
        assert per_win.max().item() <= 5e-6


# ------------------------------------------------------------------------------------------------------------------
# K7 (mlp_mfma.hip): 4096 tiles, resident dW accumulators
# ------------------------------------------------------------------------------------------------------------------
def _mlp_inputs(M, C, hid):
    x = _bf(rand((M, C), 1))
    gy = _bf(rand((M, C), 2))
    lw, lb = 1 + 0.1 * rand((C,), 3), 0.1 * rand((C,), 4)
    w1, b1 = rand((hid, C), 5, C ** -0.5), 0.1 * rand((hid,), 6)
    w2, b2 = rand((C, hid), 7, hid ** -0.5), 0.1 * rand((C,), 8)
    return x, gy, lw, lb, w1, b1, w2, b2


@pytest.mark.parametrize("C", [60, 90, 120])
def test_mlp_fused_bf16_4096_tiles(C):
    from rdst_amd import _lib
    lib = _lib.load()
    M, hid = M_FULL, 2 * C
    x, gy, lw, lb, w1, b1, w2, b2 = _mlp_inputs(M, C, hid)
    leaves = [t.clone().requires_grad_(True) for t in (x, lw, lb, w1, b1, w2, b2)]
    xr, lwr, lbr, w1r, b1r, w2r, b2r = leaves
    yref = xr + F.linear(O.gelu(F.linear(F.layer_norm(xr, (C,), lwr, lbr, 1e-5), w1r, b1r)), w2r, b2r)
    yref.backward(gy)
    gref = [t.grad for t in leaves]

    xg, gyg = x.to(DEV).bfloat16(), gy.to(DEV).bfloat16()
    P = [t.to(DEV).contiguous() for t in (lw, lb, w1, b1, w2, b2)]
    st = torch.cuda.current_stream().cuda_stream
    y = torch.full_like(xg, float("nan"))
    stats = torch.full((M, 2), float("nan"), dtype=torch.float32, device=DEV)
    nwf = lib.rdst_mlp_fwd_workspace(C, hid)
    wsf = torch.empty(max(nwf, 16), dtype=torch.uint8, device=DEV)
    _lib.check(lib.rdst_mlp_fwd(xg.data_ptr(), C, P[0].data_ptr(), P[1].data_ptr(), P[2].data_ptr(), P[3].data_ptr(),
                                P[4].data_ptr(), P[5].data_ptr(), y.data_ptr(), C, stats.data_ptr(), wsf.data_ptr(), nwf, M, C, hid, _lib.BF16,
                                st), "rdst_mlp_fwd")
    dx = torch.full_like(xg, float("nan"))
    G = [torch.full_like(t, float("nan")) for t in (P[2], P[3], P[4], P[5], P[0], P[1])]   # dW1 db1 dW2 db2 dlw dlb
    nb = lib.rdst_mlp_bwd_workspace(M, C, hid)
    wsp = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _lib.check(lib.rdst_mlp_bwd(xg.data_ptr(), C, P[0].data_ptr(), P[1].data_ptr(), stats.data_ptr(), P[2].data_ptr(),
                                P[3].data_ptr(), P[4].data_ptr(), gyg.data_ptr(), C, dx.data_ptr(), C, G[0].data_ptr(),
                                G[1].data_ptr(), G[2].data_ptr(), G[3].data_ptr(), G[4].data_ptr(), G[5].data_ptr(),
                                wsp.data_ptr(), nb, M, C, hid, _lib.BF16, st), "rdst_mlp_bwd")
    torch.cuda.synchronize()
    r = {"y": _rel(y, yref.detach()), "dx": _rel(dx, gref[0])}
    for got, want, name in zip(G, (gref[3], gref[4], gref[5], gref[6], gref[1], gref[2]),
                               ("dW1", "db1", "dW2", "db2", "dln_w", "dln_b")):
        assert torch.isfinite(got).all(), name
        r[name] = _rel(got, want)
    print(f"\nmlp bf16 C={C} M={M}: " + "  ".join(f"{k} {v:.2e}" for k, v in r.items()))
    assert all(v <= TOL for v in r.values()), r
    # tile-level: every 32-token tile of y and dx is right (loop-carried prefetch state)
    dy_t = (y.float().cpu() - yref.detach()).view(M // 32, -1).norm(dim=1) / yref.detach().view(M // 32, -1).norm(dim=1)
    dx_t = (dx.float().cpu() - gref[0]).view(M // 32, -1).norm(dim=1) / gref[0].view(M // 32, -1).norm(dim=1)
    assert dy_t.max().item() <= TOL and dx_t.max().item() <= 2 * TOL


# ------------------------------------------------------------------------------------------------------------------
# K3 forward and the one-pass Linear backward (qkv behind norm1, proj, dense tail) at M = 131072
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N,ln,res", [
    (60, 180, True, False), (90, 270, True, False), (120, 360, True, False),   # norm1 + qkv
    (60, 60, False, True), (120, 120, False, True),                             # proj + shortcut
    (120, 30, True, False),                                                     # dense tail (LN + Linear(C, 30))
])
def test_ln_linear_bf16_full_size(K, N, ln, res):
    from rdst_amd import ops
    M = M_FULL
    x = _bf(rand((B_FULL, HW * HW, K), 1))
    lw = 1 + 0.1 * rand((K,), 2) if ln else None
    lb = 0.1 * rand((K,), 3) if ln else None
    w, b = rand((N, K), 4, K ** -0.5), 0.1 * rand((N,), 5)
    r = _bf(rand((B_FULL, HW * HW, N), 6)) if res else None
    gy = _bf(rand((B_FULL, HW * HW, N), 7))

    xr = x.clone().requires_grad_(True)
    pr = [(t.clone().requires_grad_(True) if t is not None else None) for t in (lw, lb, w, b, r)]
    h = F.layer_norm(xr, (K,), pr[0], pr[1], 1e-5) if ln else xr
    yr = F.linear(h, pr[2], pr[3]) + (pr[4] if res else 0)
    yr.backward(gy)

    xg = x.to(DEV).bfloat16().requires_grad_(True)
    pg = [(t.to(DEV).requires_grad_(True) if t is not None else None) for t in (lw, lb, w, b)]
    rg = r.to(DEV).bfloat16().requires_grad_(True) if res else None
    yg = ops.ln_linear(xg, pg[0], pg[1], pg[2], pg[3], residual=rg)
    yg.backward(gy.to(DEV).bfloat16())
    torch.cuda.synchronize()
    rr = {"y": _rel(yg, yr.detach()), "dx": _rel(xg.grad, xr.grad)}
    for a, b_, n in zip(pg, pr[:4], ("dln_w", "dln_b", "dW", "db")):
        if a is not None:
            rr[n] = _rel(a.grad, b_.grad)
    print(f"\nln_linear bf16 K={K} N={N} ln={ln}: " + "  ".join(f"{k} {v:.2e}" for k, v in rr.items()))
    assert all(v <= TOL for v in rr.values()), rr
    dx_t = (xg.grad.float().cpu() - xr.grad).view(M // 32, -1).norm(dim=1) / xr.grad.view(M // 32, -1).norm(dim=1)
    assert dx_t.max().item() <= 2 * TOL


# ------------------------------------------------------------------------------------------------------------------
# K4 / K5 / K6 convs at the E1 shapes
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,res,scale,r", [
    (32, 64, 64, 150, 60, True, 1.0, 1),     # RDB fusion conv + shortcut (rdst_variations.py:444-445)
    (32, 64, 64, 60, 60, False, 1.0, 1),     # conv_after_body
    (32, 64, 64, 60, 240, False, 1.0, 2),    # upsampler stage 1 + PixelShuffle(2) (common.py:125-136)
    (8, 128, 128, 60, 240, False, 1.0, 2),   # upsampler stage 2 (a quarter of the batch: 131072 pixels)
    (8, 256, 256, 60, 1, False, 1.0, 1),     # tail conv 60 -> 1 at 256x256
])
def test_conv_bf16_full_size(B, H, W, Cin, Cout, res, scale, r):
    from rdst_amd import ops
    x = _bf(rand((B, H, W, Cin), 1))
    w = rand((Cout, Cin, 3, 3), 2, (Cin * 9) ** -0.5)
    b = 0.1 * rand((Cout,), 3)
    cy = Cout // (r * r)
    rr_ = _bf(rand((B, H * r, W * r, cy), 4)) if res else None
    gy = _bf(rand((B, H * r, W * r, cy), 5))

    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    rres = rr_.clone().requires_grad_(True) if res else None
    h = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1)
    if r > 1:
        h = F.pixel_shuffle(h, r)
    yr = h.permute(0, 2, 3, 1) * scale + (rres if res else 0)
    yr.backward(gy)

    xg = x.to(DEV).bfloat16().requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    rg = rr_.to(DEV).bfloat16().requires_grad_(True) if res else None
    yg = ops.conv_rows(xg, wg, bg, residual=rg, out_scale=scale, shuffle=r)
    yg.backward(gy.to(DEV).bfloat16())
    torch.cuda.synchronize()
    out = {"y": _rel(yg, yr.detach()), "dx": _rel(xg.grad, xr.grad), "dW": _rel(wg.grad, wr.grad),
           "db": _rel(bg.grad, br.grad)}
    print(f"\nconv bf16 {Cin}->{Cout} r={r} {B}x{H}x{W}: " + "  ".join(f"{k} {v:.2e}" for k, v in out.items()))
    assert all(v <= TOL for v in out.values()), out
    # per image row: every stripe of the output and of dx is right
    ey = (yg.float().cpu() - yr.detach()).reshape(B * H * r, -1).norm(dim=1) / yr.detach().reshape(B * H * r, -1).norm(dim=1)
    ex = (xg.grad.float().cpu() - xr.grad).reshape(B * H, -1).norm(dim=1) / xr.grad.reshape(B * H, -1).norm(dim=1)
    assert ey.max().item() <= 2 * TOL and ex.max().item() <= 2 * TOL


@pytest.mark.parametrize("B,H,W,Cin,Cout,res,scale,r", [
    (4, 64, 64, 150, 60, True, 0.7, 1),      # fusion conv: fp32 forward as two launches over halves of the input channels
    (2, 64, 64, 60, 240, False, 1.0, 2),     # upsampler conv: fp32 data gradient as two launches over halves of dY's channels
])
def test_conv_fp32_channel_slices(B, H, W, Cin, Cout, res, scale, r):
    """fp32 parity mode: a conv with more than 128 contracted channels runs on the matrix-core kernels as launches over equal
    channel slices, each after the first accumulating in place (conv_mfma.hip: conv_fwd_mfma / conv_dgrad_mfma)."""
    from rdst_amd import ops
    x = rand((B, H, W, Cin), 1)
    w = rand((Cout, Cin, 3, 3), 2, (Cin * 9) ** -0.5)
    b = 0.1 * rand((Cout,), 3)
    cy = Cout // (r * r)
    rr_ = rand((B, H * r, W * r, cy), 4) if res else None
    gy = rand((B, H * r, W * r, cy), 5)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    rres = rr_.clone().requires_grad_(True) if res else None
    h = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1)
    if r > 1:
        h = F.pixel_shuffle(h, r)
    yr = h.permute(0, 2, 3, 1) * scale + (rres if res else 0)
    yr.backward(gy)
    xg = x.to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    rg = rr_.to(DEV).requires_grad_(True) if res else None
    yg = ops.conv_rows(xg, wg, bg, residual=rg, out_scale=scale, shuffle=r)
    yg.backward(gy.to(DEV))
    torch.cuda.synchronize()
    out = {"y": _rel(yg, yr.detach()), "dx": _rel(xg.grad, xr.grad), "dW": _rel(wg.grad, wr.grad), "db": _rel(bg.grad, br.grad)}
    print(f"\nconv fp32 {Cin}->{Cout} r={r} {B}x{H}x{W}: " + "  ".join(f"{k} {v:.2e}" for k, v in out.items()))
    assert all(v <= 5e-6 for v in out.values()), out
    ey = (yg.float().cpu() - yr.detach()).reshape(B * H * r, -1).norm(dim=1) / yr.detach().reshape(B * H * r, -1).norm(dim=1)
    ex = (xg.grad.float().cpu() - xr.grad).reshape(B * H, -1).norm(dim=1) / xr.grad.reshape(B * H, -1).norm(dim=1)
    assert ey.max().item() <= 1e-5 and ex.max().item() <= 1e-5


# ------------------------------------------------------------------------------------------------------------------
# The network: RDST-E1 x4 (BASELINE.json configs[1] architecture), bf16, forward + L1 + backward vs the oracle
# ------------------------------------------------------------------------------------------------------------------
def _e1_all_gradients(B):
    from util import build_net
    cfg = O.CFG_E1
    sd = O.make_weights(cfg, 11)
    net = build_net(cfg)
    net.load_state_dict(sd, strict=True)
    net.to(DEV).train().set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(B, 1, 64, 64, generator=g)
    tgt = torch.rand(B, 1, 256, 256, generator=g)
    y = net(x.to(DEV))
    loss = F.l1_loss(y, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()

    params = dict(net.named_parameters())
    osd = {k: (v.clone().requires_grad_(True) if (k in params and params[k].requires_grad) else v) for k, v in sd.items()}
    oy = O.rdstsr_forward(x, osd, cfg)
    oloss = F.l1_loss(oy, tgt)
    oloss.backward()

    yc = y.detach().float().cpu()
    dpsnr = abs(O.psnr(tgt, yc, 4) - O.psnr(tgt, oy.detach(), 4))
    worst, tot_d, tot_r, n = (0.0, None), 0.0, 0.0, 0
    bad = []
    for k, p in params.items():
        if not p.requires_grad:
            continue
        ref = osd[k].grad
        assert p.grad is not None and ref is not None, k
        d = (p.grad.float().cpu() - ref).norm().item()
        rn = ref.norm().item()
        tot_d += d * d
        tot_r += rn * rn
        n += 1
        rel = d / max(rn, 1e-12)
        if rel > worst[0]:
            worst = (rel, k)
        if rel > 3e-2:
            bad.append((k, rel))
    total = (tot_d / tot_r) ** 0.5
    print(f"\nE1 bf16 B={B}: |dPSNR| {dpsnr:.2e} dB  out max|d| {(yc - oy.detach()).abs().max().item():.2e}  "
          f"loss {loss.item():.6f} vs {oloss.item():.6f}  {n} gradients: total rel L2 {total:.2e}, worst {worst[0]:.2e} "
          f"({worst[1]})")
    assert n == 750
    assert dpsnr < 0.05 and abs(loss.item() - oloss.item()) <= 2e-3
    assert total <= 1.2e-2
    assert not bad, bad[:10]


def test_e1_bf16_train_step_all_gradients_vs_oracle():
    """Every one of the 750 trainable tensors' gradients against the fp32 oracle (rdst_variations.py:1342-1360), batch 8
    of 64x64 (512 windows: every persistent workgroup loops).  bf16 activations through 48 Swin blocks: stated tolerance
    rel L2 <= 3e-2 per tensor (most sit near 1e-2), total gradient rel L2 <= 1.2e-2, |dPSNR| < 0.05 dB (printed)."""
    _e1_all_gradients(8)


def test_e1_bf16_train_step_all_gradients_vs_oracle_bench_batch():
    """The same at THE batch bench.py runs (BASELINE configs[1]: B = 32, 2048 windows per Swin block)."""
    _e1_all_gradients(32)


def _ws16_all_gradients(dtype, B, tol_tensor, tol_total, tol_dpsnr, tol_loss):
    from util import build_net
    cfg = O.CFG_WS16
    sd = O.make_weights(cfg, 12)
    net = build_net(cfg)
    net.load_state_dict(sd, strict=True)
    net.to(DEV).train().set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(4321)
    x = torch.rand(B, 3, 128, 128, generator=g)
    tgt = torch.rand(B, 3, 256, 256, generator=g)
    y = net(x.to(DEV))
    loss = F.l1_loss(y, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()

    params = dict(net.named_parameters())
    osd = {k: (v.clone().requires_grad_(True) if (k in params and params[k].requires_grad) else v) for k, v in sd.items()}
    oy = O.rdstsr_forward(x, osd, cfg)
    oloss = F.l1_loss(oy, tgt)
    oloss.backward()

    yc = y.detach().float().cpu()
    dpsnr = abs(O.psnr(tgt, yc, 2) - O.psnr(tgt, oy.detach(), 2))
    worst, tot_d, tot_r, n, bad = (0.0, None), 0.0, 0.0, 0, []
    for k, p in params.items():
        if not p.requires_grad:
            continue
        ref = osd[k].grad
        assert p.grad is not None and ref is not None, k
        d = (p.grad.float().cpu() - ref).norm().item()
        rn = ref.norm().item()
        tot_d += d * d
        tot_r += rn * rn
        n += 1
        rel = d / max(rn, 1e-12)
        if rel > worst[0]:
            worst = (rel, k)
        if rel > tol_tensor:
            bad.append((k, rel))
    total = (tot_d / tot_r) ** 0.5
    print(f"\nws16 {str(dtype).split('.')[-1]} B={B}: |dPSNR| {dpsnr:.2e} dB  out max|d| {(yc - oy.detach()).abs().max().item():.2e}  "
          f"loss {loss.item():.6f} vs {oloss.item():.6f}  {n} gradients: total rel L2 {total:.2e}, worst {worst[0]:.2e} "
          f"({worst[1]})")
    assert n == sum(1 for v in osd.values() if v.requires_grad) == 748   # one PixelShuffle stage (x2): two tensors fewer than E1
    assert dpsnr < tol_dpsnr and abs(loss.item() - oloss.item()) <= tol_loss
    assert total <= tol_total
    assert not bad, bad[:10]


def test_ws16_bf16_train_step_all_gradients_vs_oracle():
    """BASELINE configs[3] (3-channel x2, 128x128 -> 256x256, window 16, the E1 widths) in the bf16 mode: every trainable
    tensor's gradient against the fp32 oracle, batch 2 (128 windows per layer: wattn16_mfma.hip forward and backward in all
    48 Swin blocks, shifted and not, with the region masks of the last window row / column).  Same stated tolerances as the
    E1 test except per tensor: rel L2 <= 4.5e-2 (B = 2 is 8192 tokens, a sixteenth of the E1 test's: the 961-entry bias tables
    sit at 3.0e-2..3.7e-2, measured), total <= 1.2e-2 (measured 4.8e-3), |dPSNR| < 0.05 dB."""
    _ws16_all_gradients(torch.bfloat16, 2, 4.5e-2, 1.2e-2, 0.05, 2e-3)


def test_ws16_fp32_train_step_all_gradients_vs_oracle():
    """BASELINE configs[3] in the reference's own arithmetic: the full window-16 network (48 Swin blocks on the exact-fp32
    matrix-core kernels of wattn16_f32.hip, every mask case of the shifted blocks) in fp32, forward + L1 + backward at the
    configuration's own patch size (B = 1: 64 windows per layer), every trainable tensor's gradient against the oracle:
    rel L2 <= 1e-4 per tensor (measured worst 3.4e-6), <= 1e-5 in total (2.5e-7), |dPSNR| < 5e-6 dB (5.1e-8; the claim is four
    decimals), loss to 1e-6."""
    _ws16_all_gradients(torch.float32, 1, 1e-4, 1e-5, 5e-6, 1e-6)


def test_e1_fp32_bench_shape_psnr_equal_to_4_decimals():
    """north_star's stated tolerance at THE benchmark shape: RDST-E1 x4 on 1x64x64 LR patches (B = 4 of BASELINE configs[1]'s
    32: the oracle is a CPU pass), parity mode (fp32 activations, exact-fp32 matrix-core kernels), forward + L1 + backward
    against the oracle: |dPSNR| < 5e-5 dB (PSNR equal to >= 4 decimal places, metrics/sr_metrics.py:8-9, border 4 as
    trans_sr_tester.py:155 passes it), loss to 1e-6, every one of the 750 gradients to 1e-3 relative."""
    from util import build_net
    cfg = O.CFG_E1
    B = 4
    sd = O.make_weights(cfg, 11)
    net = build_net(cfg)
    net.load_state_dict(sd, strict=True)
    net.to(DEV).train()
    g = torch.Generator().manual_seed(4321)
    x = torch.rand(B, 1, 64, 64, generator=g)
    tgt = torch.rand(B, 1, 256, 256, generator=g)
    y = net(x.to(DEV))
    loss = F.l1_loss(y, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    osd = {k: (v.clone().requires_grad_(True) if (k in params and params[k].requires_grad) else v) for k, v in sd.items()}
    oy = O.rdstsr_forward(x, osd, cfg)
    oloss = F.l1_loss(oy, tgt)
    oloss.backward()
    yc = y.detach().float().cpu()
    p_hip, p_ref = O.psnr(tgt, yc, 4), O.psnr(tgt, oy.detach(), 4)
    worst = max(((p.grad.cpu() - osd[k].grad).norm().item() / max(osd[k].grad.norm().item(), 1e-12), k)
                for k, p in params.items() if p.requires_grad)
    print(f"\nE1 fp32 B={B} 64x64: PSNR {p_hip:.6f} vs {p_ref:.6f} dB (|d| {abs(p_hip - p_ref):.2e})  out max|d| "
          f"{(yc - oy.detach()).abs().max().item():.2e}  loss {loss.item():.7f} vs {oloss.item():.7f}  worst gradient {worst[0]:.2e} ({worst[1]})")
    assert abs(p_hip - p_ref) < 5e-5
    assert abs(loss.item() - oloss.item()) <= 1e-6
    assert worst[0] <= 1e-3, worst
