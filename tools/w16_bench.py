"""Window-16 attention (wattn16_mfma.hip) at the ws16 bench shape (B = 8 of 128 x 128, 6 heads, bf16): us per launch,
forward and backward.   python tools/w16_bench.py [C ...]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W, heads, ws = 8, 128, 128, 6, 16
bf = torch.bfloat16


def timed(run, n=10):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for C in [int(c) for c in sys.argv[1:]] or [60, 90, 120]:
    q = torch.randn(B, H, W, 3 * C, device=dev).to(bf)
    g = torch.randn(B, H, W, C, device=dev).to(bf)
    o = torch.empty(B, H, W, C, device=dev, dtype=bf)
    d = torch.empty_like(q)
    table = 0.5 * torch.randn(961, heads, device=dev)
    dtab = torch.zeros(961, heads, device=dev)
    nws = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
    wsp = torch.empty(nws, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    sc = (C // heads) ** -0.5
    def fwd():
        _lib.check(lib.rdst_wattn_fwd(q.data_ptr(), 3 * C, table.data_ptr(), None, 0, o.data_ptr(), C, B, H, W, C, heads, ws, 8,
                                      sc, _lib.BF16, st), "rdst_wattn_fwd")
    def bwd():
        _lib.check(lib.rdst_wattn_bwd(q.data_ptr(), 3 * C, table.data_ptr(), None, 0, g.data_ptr(), C, d.data_ptr(), 3 * C,
                                      dtab.data_ptr(), wsp.data_ptr(), nws, B, H, W, C, heads, ws, 8, sc, _lib.BF16, st),
                   "rdst_wattn_bwd")
    nlse = torch.empty(B * H * W, heads, device=dev)
    def fwd2():
        _lib.check(lib.rdst_wattn_fwd_lse(q.data_ptr(), 3 * C, table.data_ptr(), o.data_ptr(), C, nlse.data_ptr(), B, H, W, C, heads,
                                          ws, 8, sc, _lib.BF16, st), "rdst_wattn_fwd_lse")
    def bwd2():
        _lib.check(lib.rdst_wattn_bwd_lse(q.data_ptr(), 3 * C, table.data_ptr(), g.data_ptr(), C, o.data_ptr(), C, nlse.data_ptr(),
                                          d.data_ptr(), 3 * C, dtab.data_ptr(), wsp.data_ptr(), nws, B, H, W, C, heads, ws, 8, sc,
                                          _lib.BF16, st), "rdst_wattn_bwd_lse")
    print(f"C={C}: fwd {timed(fwd):7.1f} us   bwd {timed(bwd):7.1f} us   |  with statistics: fwd {timed(fwd2):7.1f} us   bwd {timed(bwd2):7.1f} us", flush=True)
