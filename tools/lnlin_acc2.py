"""rdst_ln_linear_bwd2 (qkv shape, N = 3K) at M = 131072: contiguous x / strided x (dense-buffer view) / + second addend.
usage: python tools/lnlin_acc2.py [K ...]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
M = 131072
for K in [int(c) for c in sys.argv[1:]] or [60, 90, 120]:
    N, Wd = 3 * K, K + 90
    NB = 4
    wide = [torch.randn(M, Wd, device=dev).bfloat16() for _ in range(NB)]
    gwide = [torch.randn(M, Wd, device=dev).bfloat16() for _ in range(NB)]
    xc = [w[:, :K].contiguous() for w in wide]
    dy = [torch.randn(M, N, device=dev).bfloat16() for _ in range(NB)]
    add = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NB)]
    dx = [torch.empty(M, K, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    w = torch.randn(N, K, device=dev) * K ** -0.5
    lw, lb = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    stats = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], dim=1).contiguous()
    G = [torch.empty_like(w), torch.empty(N, device=dev), torch.empty(K, device=dev), torch.empty(K, device=dev)]
    nws = lib.rdst_ln_linear_bwd_workspace(M, K, N)
    wsp = [torch.empty(nws, dtype=torch.uint8, device=dev) for _ in range(NB)]
    for name, xs, ldx, a2 in (("x contiguous", xc, K, False), ("x strided", [t for t in wide], Wd, False), ("x strided + addend2", wide, Wd, True)):
        def run(i, st):
            _lib.check(lib.rdst_ln_linear_bwd2(xs[i].data_ptr(), ldx, lw.data_ptr(), lb.data_ptr(), stats.data_ptr(), 0, w.data_ptr(),
                                               dy[i].data_ptr(), N, dx[i].data_ptr(), K, add[i].data_ptr(), K, G[0].data_ptr(), G[1].data_ptr(),
                                               G[2].data_ptr(), G[3].data_ptr(), wsp[i].data_ptr(), nws, M, K, N, 1.0, _lib.BF16, st,
                                               gwide[i].data_ptr() if a2 else None, Wd), "bwd2")
        run(0, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.rdst_reduce_batch_begin(), "b")
            for i in range(NB):
                run(i, st)
            _lib.check(lib.rdst_reduce_batch_end(st), "e")
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"K={K:3d} {name:22s}: {1e3 * e0.elapsed_time(e1) / (5 * NB):7.2f} us per call (+ 1/{NB} of the batched reductions)", flush=True)
