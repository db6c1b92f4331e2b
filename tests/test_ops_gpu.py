"""K3/K4/K5 parity: fused LayerNorm/Linear and conv ops (through the C ABI) vs plain torch fp32 on CPU."""
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O
from util import rand

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _leaf(t):
    return t.clone().requires_grad_(True)


def _gpu(t, dtype=None):
    t = t.to(DEV)
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(True)


LIN = [  # M, K, N, ln, act, residual, scale
    (256, 60, 180, True, 0, False, 1.0),     # norm1 + qkv
    (256, 60, 60, False, 0, True, 1.0),      # proj + shortcut
    (200, 90, 180, True, 0, False, 1.0),     # norm2 + fc1 (M not a tile multiple)
    (256, 180, 90, False, 1, True, 1.0),     # GELU + fc2 + residual
    (192, 120, 30, True, 0, False, 0.5),     # dense tail, dense_scale
    (288, 120, 30, True, 0, False, 1.0),     # dense tail as E1 runs it (one-pass backward, 1 n-tile on 8 waves)
    (160, 90, 270, True, 0, False, 1.0),     # norm1 + qkv at C = 90 (one-pass backward, 9 waves)
    (130, 48, 48, True, 0, True, 0.9),       # LayerNorm only (no weight), scale + residual
    (4133, 60, 60, True, 0, True, 0.9),      # ... at E1's width (the last 8-channel chunk of a row overlaps its neighbour), many row groups
    (64, 33, 17, False, 2, False, 1.0),      # odd sizes, LeakyReLU input
    (384, 90, 270, True, 0, False, 1.0),     # fp32: the weights fit the LDS only as ONE chunk that ends inside a column tile (launch_lin)
    (320, 240, 120, False, 1, True, 1.0),    # GELU + fc2 at C = 120 (fp32: the same single tight chunk), + residual
    (416, 120, 240, True, 0, False, 1.0),    # norm2 + fc1 at C = 120 (fp32: tight chunk that is a whole number of tiles)
]


@pytest.mark.parametrize("M,K,N,ln,act,res,scale", LIN)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ln_linear(M, K, N, ln, act, res, scale, dtype):
    from rdst_amd import ops
    ln_only = ln and K == N and res and scale == 0.9
    x = rand((2, M // 2, K), 1)
    lw = 1 + 0.1 * rand((K,), 2) if ln else None
    lb = 0.1 * rand((K,), 3) if ln else None
    w = None if ln_only else rand((N, K), 4, K ** -0.5)
    b = None if ln_only else 0.1 * rand((N,), 5)
    r = rand((2, M // 2, N), 6) if res else None
    gy = rand((2, M // 2, N), 7)
    if dtype == torch.bfloat16:  # identical (bf16-representable) inputs on both sides
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
        r = r.bfloat16().float() if res else None

    xr = _leaf(x)
    pr = [(_leaf(t) if t is not None else None) for t in (lw, lb, w, b, r)]
    h = xr
    if ln:
        h = F.layer_norm(h, (K,), pr[0], pr[1], 1e-5)
    elif act == 1:
        h = O.gelu(h)
    elif act == 2:
        h = F.leaky_relu(h, 0.2)
    if pr[2] is not None:
        h = F.linear(h, pr[2], pr[3])
    yr = h * scale + (pr[4] if res else 0)
    yr.backward(gy)

    xg = _gpu(x, dtype)
    pg = [(_gpu(t) if t is not None else None) for t in (lw, lb, w, b)]
    rg = _gpu(r, dtype) if res else None
    yg = ops.ln_linear(xg, pg[0], pg[1], pg[2], pg[3], in_act=act, residual=rg, out_scale=scale)
    yg.backward(gy.to(DEV).to(dtype))
    torch.cuda.synchronize()

    tol = 3e-5 if dtype == torch.float32 else 3e-2
    assert (yg.float().cpu() - yr).abs().max().item() <= tol * max(1.0, yr.abs().max().item())
    gtol = 1e-4 if dtype == torch.float32 else 2e-2

    def rel(a, b_):
        return (a.float().cpu() - b_).norm().item() / max(b_.norm().item(), 1e-12)
    assert rel(xg.grad, xr.grad) <= gtol
    for a, b_ in zip(pg, pr[:4]):
        if a is not None:
            assert rel(a.grad, b_.grad) <= gtol
    if res:
        assert rel(rg.grad, pr[4].grad) <= gtol


CONV = [  # B, H, W, Cin, Cout, k, act, residual, scale, shuffle
    (2, 8, 8, 150, 60, 3, 0, True, 0.7, 1),    # RDB fusion conv + residual_scale + shortcut
    (1, 8, 12, 60, 240, 3, 0, False, 1.0, 2),  # upsampler conv + PixelShuffle(2), non-square
    (2, 9, 7, 1, 60, 3, 0, False, 1.0, 1),     # head conv, odd sizes
    (1, 16, 16, 60, 1, 3, 0, False, 1.0, 1),   # last conv
    (2, 24, 40, 60, 1, 3, 0, False, 0.5, 1),   # last conv: several workgroups, ragged tail, scale
    (1, 9, 11, 36, 1, 3, 0, False, 1.0, 1),    # one output channel, channel count not a multiple of 8
    (2, 6, 6, 37, 12, 1, 2, False, 1.0, 1),    # 1x1 conv reading through LeakyReLU ('3conv')
    (1, 6, 6, 48, 108, 3, 0, False, 1.0, 3),   # x3 upsampler
    (2, 5, 5, 3, 3, 1, 0, False, 1.0, 1),      # MeanShift
    (2, 16, 25, 1, 1, 1, 0, True, 0.5, 1),     # MeanShift on a single-channel image (elementwise kernel), + residual, ragged tail
    (1, 7, 9, 1, 1, 1, 0, True, 1.0, 1),       # ... 63 pixels: the last thread's chunk is cut (fp32: 4 per thread, bf16: 8)
    (1, 8, 32, 150, 60, 3, 0, True, 0.7, 1),   # fusion conv on full 32-pixel row slabs (fp32: two launches over halves of the input channels)
    (1, 4, 32, 60, 240, 3, 0, False, 1.0, 2),  # upsampler conv, 32-pixel row slabs (fp32 data gradient: two launches over halves of dY's channels)
    (1, 6, 10, 150, 60, 3, 2, True, 1.0, 1),   # the same split on the pixel-tile kernel, reading through LeakyReLU
]


@pytest.mark.parametrize("B,H,W,Cout,scale", [(2, 9, 7, 60, 1.0), (3, 40, 70, 60, 0.5), (1, 16, 16, 36, 1.0)])
def test_head_conv_one_input_channel(B, H, W, Cout, scale):
    """The head conv of RDSTSR on a single-channel image (rdst_variations.py:1213, default_conv(1, embed_dim, 3)), bf16: its
    input is the (mean-shifted) image, which needs no gradient — forward and weight gradient run on the tile kernels of
    conv_c1.hip in their mirrored form (tiles ragged in both directions, several tiles per workgroup chunk, a channel count that
    is not a multiple of 8)."""
    from rdst_amd import ops
    x = rand((B, H, W, 1), 1).bfloat16().float()
    w = rand((Cout, 1, 3, 3), 2, 1 / 3.0)
    b = 0.1 * rand((Cout,), 3)
    gy = rand((B, H, W, Cout), 5).bfloat16().float()
    wr, br = _leaf(w), _leaf(b)
    yr = F.conv2d(x.permute(0, 3, 1, 2), wr, br, padding=1).permute(0, 2, 3, 1) * scale
    yr.backward(gy)
    xg, wg, bg = x.to(DEV).bfloat16(), _gpu(w), _gpu(b)     # xg: no gradient requested
    yg = ops.conv_rows(xg, wg, bg, out_scale=scale)
    yg.backward(gy.to(DEV).bfloat16())
    torch.cuda.synchronize()
    assert (yg.float().cpu() - yr.detach()).abs().max().item() <= 2e-2 * max(1.0, yr.abs().max().item())

    def rel(a, b_):
        return (a.float().cpu() - b_).norm().item() / max(b_.norm().item(), 1e-12)
    assert rel(wg.grad, wr.grad) <= 1e-2 and rel(bg.grad, br.grad) <= 1e-2


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,act,res,scale,r", CONV)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_rows(B, H, W, Cin, Cout, k, act, res, scale, r, dtype):
    from rdst_amd import ops
    x = rand((B, H, W, Cin), 1)
    w = rand((Cout, Cin, k, k), 2, (Cin * k * k) ** -0.5)
    b = 0.1 * rand((Cout,), 3)
    cy = Cout // (r * r)
    rr = rand((B, H * r, W * r, cy), 4) if res else None
    gy = rand((B, H * r, W * r, cy), 5)
    if dtype == torch.bfloat16:
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
        rr = rr.bfloat16().float() if res else None

    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    rres = _leaf(rr) if res else None
    h = xr.permute(0, 3, 1, 2)
    if act == 2:
        h = F.leaky_relu(h, 0.2)
    h = F.conv2d(h, wr, br, padding=k // 2)
    if r > 1:
        h = F.pixel_shuffle(h, r)
    yr = h.permute(0, 2, 3, 1) * scale + (rres if res else 0)
    yr.backward(gy)

    xg, wg, bg = _gpu(x, dtype), _gpu(w), _gpu(b)
    rg = _gpu(rr, dtype) if res else None
    yg = ops.conv_rows(xg, wg, bg, in_act=act, residual=rg, out_scale=scale, shuffle=r)
    yg.backward(gy.to(DEV).to(dtype))
    torch.cuda.synchronize()

    tol = 3e-5 if dtype == torch.float32 else 3e-2
    assert (yg.float().cpu() - yr).abs().max().item() <= tol * max(1.0, yr.abs().max().item())
    gtol = 1e-4 if dtype == torch.float32 else 2e-2

    def rel(a, b_):
        return (a.float().cpu() - b_).norm().item() / max(b_.norm().item(), 1e-12)
    assert rel(xg.grad, xr.grad) <= gtol
    assert rel(wg.grad, wr.grad) <= gtol
    assert rel(bg.grad, br.grad) <= gtol
    if res:
        assert rel(rg.grad, rres.grad) <= gtol


def test_layout_roundtrip():
    from rdst_amd import ops
    x = rand((2, 3, 5, 7), 9).to(DEV)
    rows = ops.nchw_to_rows(x, torch.float32)
    assert torch.equal(rows, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.rows_to_nchw(rows), x)
    assert torch.equal(ops.rows_to_nchw(ops.nchw_to_rows(x, torch.bfloat16)), x.bfloat16().float())


@pytest.mark.parametrize("M,K,N,ln", [(160, 120, 360, True), (96, 60, 60, False), (200, 90, 90, False), (64, 48, 20, True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ln_linear_bwd_dx_add(M, K, N, ln, dtype):
    """rdst_ln_linear_bwd with dX_add (the residual fan-out sum folded into the kernel): dX = dX_add + f'(...),
    called through the C ABI; covers the one-pass backward kernels (bf16) and the composed path (fp32, small N)."""
    from rdst_amd import _lib
    lib = _lib.load()
    x = rand((M, K), 11)
    gy = rand((M, N), 12)
    add = rand((M, K), 13)
    lw, lb = (1 + 0.1 * rand((K,), 14), 0.1 * rand((K,), 15)) if ln else (None, None)
    w, b = rand((N, K), 16, K ** -0.5), 0.1 * rand((N,), 17)
    if dtype == torch.bfloat16:
        x, gy, add = x.bfloat16().float(), gy.bfloat16().float(), add.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) if t is not None else None for t in (lw, lb, w, b)]
    h = F.layer_norm(xr, (K,), pr[0], pr[1], 1e-5) if ln else xr
    F.linear(h, pr[2], pr[3]).backward(gy)
    want_dx = xr.grad + add

    code = _lib.F32 if dtype == torch.float32 else _lib.BF16
    xg, gyg, addg = x.to(DEV).to(dtype), gy.to(DEV).to(dtype), add.to(DEV).to(dtype)
    P = [t.to(DEV).contiguous() if t is not None else None for t in (lw, lb, w, b)]
    stats = None
    if ln:
        xf = xg.float()
        stats = torch.stack([xf.mean(-1), (xf.var(-1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    dx = torch.empty_like(xg)
    dW, db = torch.empty_like(P[2]), torch.empty_like(P[3])
    dlw = torch.empty_like(P[0]) if ln else None
    dlb = torch.empty_like(P[1]) if ln else None
    nb = lib.rdst_ln_linear_bwd_workspace(M, K, N)
    wsp = torch.empty(nb, dtype=torch.uint8, device=DEV)

    def ptr(t):
        return None if t is None else t.data_ptr()
    _lib.check(lib.rdst_ln_linear_bwd(xg.data_ptr(), K, ptr(P[0]), ptr(P[1]), ptr(stats), 0, P[2].data_ptr(), gyg.data_ptr(), N,
                                      dx.data_ptr(), K, addg.data_ptr(), K, dW.data_ptr(), db.data_ptr(), ptr(dlw), ptr(dlb),
                                      wsp.data_ptr(), nb, M, K, N, 1.0, code, torch.cuda.current_stream().cuda_stream),
               "rdst_ln_linear_bwd")
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == torch.float32 else 2e-2

    def rel(a, b_):
        return (a.float().cpu() - b_).norm().item() / max(b_.norm().item(), 1e-12)
    assert rel(dx, want_dx) <= tol
    assert rel(dW, pr[2].grad) <= tol and rel(db, pr[3].grad) <= tol
    if ln:
        assert rel(dlw, pr[0].grad) <= tol and rel(dlb, pr[1].grad) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [(2, 5, 7, 48), (1, 16, 16, 64), (3, 4, 6, 61)])
def test_upsample_nearest2_vs_torch(B, H, W, C, dtype):
    """rdst_upsample2_fwd / _bwd (csrc/upsample.hip) vs F.interpolate(scale_factor=2, mode='nearest') and its autograd
    (swin_transformer_sr.py:801-802): the forward is a copy (bit-exact), the backward a 4-term sum (fp32: exact to rounding
    order, bf16: one rounding of the fp32 sum)."""
    from rdst_amd import ops
    x = rand((B, H, W, C), 31).to(dtype)
    gy = rand((B, 2 * H, 2 * W, C), 32).to(dtype)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="nearest")
    yr.backward(gy.float().permute(0, 3, 1, 2))
    xg = x.to(DEV).requires_grad_(True)
    y = ops.upsample_nearest2(xg)
    y.backward(gy.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(y.detach().float().cpu(), yr.detach().permute(0, 2, 3, 1))
    want = xr.grad.permute(0, 2, 3, 1)
    if dtype == torch.float32:
        assert (xg.grad.cpu() - want).abs().max().item() <= 1e-6
    else:
        assert torch.equal(xg.grad.cpu(), want.bfloat16())
    # a strided source (rows inside a wider buffer) gives the same result
    wide = torch.zeros(B, H, W, C + 8, dtype=dtype, device=DEV)
    wide[..., 4:4 + C] = x.to(DEV)
    assert torch.equal(ops.upsample_nearest2(wide[..., 4:4 + C]), y.detach())
