"""Cold timing of the fused Mlp forward (rdst_mlp_fwd, bf16, prepacked-size workspace) at M = 131072 in one HIP graph.
usage: python tools/mlp_fwd_cold.py [C ...]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
M, NBUF = 131072, 8
for C in [int(c) for c in sys.argv[1:]] or [60, 90, 120]:
    hid = 2 * C
    X = [torch.randn(M, C, device=dev).bfloat16() for _ in range(NBUF)]
    Y = [torch.empty(M, C, device=dev, dtype=torch.bfloat16) for _ in range(NBUF)]
    stats = torch.empty(M, 2, device=dev)
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = torch.randn(hid, C, device=dev) * C ** -0.5, torch.zeros(hid, device=dev)
    w2, b2 = torch.randn(C, hid, device=dev) * hid ** -0.5, torch.zeros(C, device=dev)
    nws = lib.rdst_mlp_fwd_workspace(C, hid)
    wsp = torch.empty(nws, dtype=torch.uint8, device=dev)
    def run(i, st, nb):
        _lib.check(lib.rdst_mlp_fwd(X[i].data_ptr(), C, lw.data_ptr(), lb.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                    b2.data_ptr(), Y[i].data_ptr(), C, stats.data_ptr(), wsp.data_ptr(), nb, M, C, hid, _lib.BF16, st), "mlp_fwd")
    run(0, torch.cuda.current_stream().cuda_stream, nws)      # packs the images
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(NBUF):
            run(i, torch.cuda.current_stream().cuda_stream, _lib.PREPACKED)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C:3d}: {1e3 * e0.elapsed_time(e1) / (5 * NBUF):7.2f} us per call", flush=True)
