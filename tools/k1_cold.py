"""Cold timing of the window-attention forward (K1) at the bench shape: NBUF distinct qkv buffers (more than the
256 MiB Infinity Cache holds), all launches captured into ONE HIP graph and replayed inside one event pair.
usage: python tools/k1_cold.py [C ...]   (env switches of the debug library apply: RDST_HIP_LIB=..., RDST_K1_MINB=4)"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import ops

dev = torch.device("cuda:0")
B, H, W, heads, ws = 32, 64, 64, 6, 8
Cs = [int(c) for c in sys.argv[1:]] or [60, 90, 120]
for C in Cs:
    nbuf = max(8, int(600e6 / (B * H * W * 3 * C * 2)) + 1)
    qs = [torch.randn(B, H, W, 3 * C, device=dev).bfloat16() for _ in range(nbuf)]
    table = 0.5 * torch.randn(225, heads, device=dev)
    scale = (C // heads) ** -0.5
    for shift in (0, 4):
        with torch.no_grad():
            for q in qs[:2]:
                ops.window_attention(q, table, H, W, heads, ws, shift, scale)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = [ops.window_attention(q, table, H, W, heads, ws, shift, scale) for q in qs]
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (reps * nbuf)
        nb = B * H * W * 4 * C * 2
        print(f"C={C:3d} shift={shift} cold x{nbuf}: {us:7.2f} us  {nb / us / 1e3:7.1f} GB/s  frac {nb / us / 1e3 / 8000:.3f}", flush=True)
        del outs, g
    del qs
