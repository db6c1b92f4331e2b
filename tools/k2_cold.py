"""Cold timing of the window-attention backward (K2) at the bench shape through the C ABI: NBUF distinct
(qkv, dout, dqkv) sets, all launches captured into ONE HIP graph, replayed inside one event pair.
usage: python tools/k2_cold.py [C ...]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W, heads, ws = 32, 64, 64, 6, 8
Cs = [int(c) for c in sys.argv[1:]] or [60, 90, 120]
for C in Cs:
    nbuf = 6
    qs = [torch.randn(B, H, W, 3 * C, device=dev).bfloat16() for _ in range(nbuf)]
    gs = [torch.randn(B, H, W, C, device=dev).bfloat16() for _ in range(nbuf)]
    ds = [torch.empty(B, H, W, 3 * C, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
    table = 0.5 * torch.randn(225, heads, device=dev)
    dtab = torch.zeros(225, heads, device=dev)
    nws = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
    wsp = torch.empty(nws, dtype=torch.uint8, device=dev)
    scale = (C // heads) ** -0.5
    for shift in (0, 4):
        def run(i, st):
            _lib.check(lib.rdst_wattn_bwd(qs[i].data_ptr(), 3 * C, table.data_ptr(), None, 0, gs[i].data_ptr(), C,
                                          ds[i].data_ptr(), 3 * C, dtab.data_ptr(), wsp.data_ptr(), nws, B, H, W, C, heads,
                                          ws, shift, scale, _lib.BF16, st), "rdst_wattn_bwd")
        run(0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nbuf):
                run(i, torch.cuda.current_stream().cuda_stream)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (reps * nbuf)
        nb = B * H * W * 7 * C * 2
        print(f"C={C:3d} shift={shift} K2 + table reduce, cold x{nbuf}: {us:7.2f} us  {nb / us / 1e3:7.1f} GB/s  frac {nb / us / 1e3 / 8000:.3f}", flush=True)
        del g
