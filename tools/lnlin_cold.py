"""Cold timing of the one-pass Linear backward (rdst_ln_linear_bwd, bf16) at M = 131072 for the nine E1 shapes:
NBUF operand sets, all calls of a shape captured into ONE HIP graph inside a reduce batch (the kernel alone: the slab
sums are deferred to one batched launch at the end of the graph), replayed inside one event pair.
usage: python tools/lnlin_cold.py [K N ln ...]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
import os
M, NBUF = int(os.environ.get("LNM", "131072")), 6
SHAPES = [(60, 180, 1), (90, 270, 1), (120, 360, 1), (60, 60, 0), (90, 90, 0), (120, 120, 0), (60, 30, 1), (90, 30, 1), (120, 30, 1)]
a = [int(v) for v in sys.argv[1:]]
if a:
    SHAPES = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for K, N, ln in SHAPES:
    X = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NBUF)]
    dY = [torch.randn(M, N, device=dev).bfloat16() for _ in range(NBUF)]
    dX = [torch.empty(M, K, device=dev, dtype=torch.bfloat16) for _ in range(NBUF)]
    acc = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NBUF)] if N == 3 * K else [None] * NBUF
    stats = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], dim=1).contiguous()
    W = torch.randn(N, K, device=dev) * K ** -0.5
    lw, lb = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    dW, db, dlw, dlb = torch.empty_like(W), torch.empty(N, device=dev), torch.empty(K, device=dev), torch.empty(K, device=dev)
    nws = lib.rdst_ln_linear_bwd_workspace(M, K, N)
    wsp = [torch.empty(nws, dtype=torch.uint8, device=dev) for _ in range(NBUF)]

    def run(i, st):
        _lib.check(lib.rdst_ln_linear_bwd(X[i].data_ptr(), K, lw.data_ptr() if ln else None, lb.data_ptr() if ln else None,
                                          stats.data_ptr() if ln else None, 0, W.data_ptr(), dY[i].data_ptr(), N, dX[i].data_ptr(), K,
                                          acc[i].data_ptr() if acc[i] is not None else None, K, dW.data_ptr(), db.data_ptr(),
                                          dlw.data_ptr() if ln else None, dlb.data_ptr() if ln else None, wsp[i].data_ptr(), nws,
                                          M, K, N, 1.0, _lib.BF16, st), "rdst_ln_linear_bwd")
    st = torch.cuda.current_stream().cuda_stream
    run(0, st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.rdst_reduce_batch_begin(), "begin")
        for i in range(NBUF):
            run(i, st)
        _lib.check(lib.rdst_reduce_batch_end(st), "end")
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / (reps * NBUF)
    nb = M * 2 * (2 * K + N + (K if acc[0] is not None else 0))
    print(f"K={K:3d} N={N:3d} ln={ln}: {us:7.2f} us per call (kernel + 1/{NBUF} of the batched sums)  {nb / us / 1e3:7.1f} GB/s  frac {nb / us / 1e3 / 8000:.3f}", flush=True)
