"""Cold timing of the fused Mlp backward (rdst_mlp_bwd, bf16) at M = 131072: NBUF operand sets captured into ONE HIP graph
inside a reduce batch, replayed inside one event pair.  usage: python tools/mlp_bwd_cold.py [C ...]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
M, NBUF = 131072, 6
for C in [int(c) for c in sys.argv[1:]] or [60, 90, 120]:
    hid = 2 * C
    X = [torch.randn(M, C, device=dev).bfloat16() for _ in range(NBUF)]
    dY = [torch.randn(M, C, device=dev).bfloat16() for _ in range(NBUF)]
    dX = [torch.empty(M, C, device=dev, dtype=torch.bfloat16) for _ in range(NBUF)]
    stats = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], dim=1).contiguous()
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = torch.randn(hid, C, device=dev) * C ** -0.5, torch.zeros(hid, device=dev)
    w2 = torch.randn(C, hid, device=dev) * hid ** -0.5
    G = [torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty(C, device=dev), torch.empty_like(lw), torch.empty_like(lb)]
    nb = lib.rdst_mlp_bwd_workspace(M, C, hid)
    wsp = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(NBUF)]

    def run(i, st):
        _lib.check(lib.rdst_mlp_bwd(X[i].data_ptr(), C, lw.data_ptr(), lb.data_ptr(), stats.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                    w2.data_ptr(), dY[i].data_ptr(), C, dX[i].data_ptr(), C, G[0].data_ptr(), G[1].data_ptr(),
                                    G[2].data_ptr(), G[3].data_ptr(), G[4].data_ptr(), G[5].data_ptr(), wsp[i].data_ptr(), nb, M, C,
                                    hid, _lib.BF16, st), "rdst_mlp_bwd")
    run(0, torch.cuda.current_stream().cuda_stream)
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
    print(f"C={C:3d}: {us:7.2f} us per call (kernels + 1/{NBUF} of the batched sums)", flush=True)
