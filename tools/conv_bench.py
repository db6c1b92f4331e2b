#!/usr/bin/env python3
"""Micro-benchmark of the conv op (forward and backward) at the E1 shapes: average kernel time over a HIP-graph replay
of REPS back-to-back calls on rotating buffers, and the fraction of the bf16 MFMA peak (2.5 PFLOP/s) it corresponds to."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"     # python tools/conv_bench.py [bf16 | fp32 | fp32x3]
dt = torch.bfloat16 if mode == "bf16" else torch.float32
ops.set_f32_split(mode == "fp32x3")
SHAPES = [(32, 64, 64, 150, 60, 1, True), (32, 64, 64, 60, 60, 1, False), (32, 64, 64, 60, 240, 2, False),
          (32, 128, 128, 60, 240, 2, False)]
REPS = 8


def timed(fn, n=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, H, W, cin, cout, r, res in SHAPES:
    w = (torch.randn(cout, cin, 3, 3, device=dev) * (cin * 9) ** -0.5).requires_grad_(True)
    b = torch.zeros(cout, device=dev, requires_grad=True)
    xs = [torch.randn(B, H, W, cin, device=dev).to(dt).requires_grad_(True) for _ in range(4)]
    cy = cout // (r * r)
    rs = [torch.randn(B, H * r, W * r, cy, device=dev).to(dt) for _ in range(4)] if res else [None] * 4
    gy = torch.randn(B, H * r, W * r, cy, device=dev).to(dt)
    flop = 2.0 * B * H * W * cin * cout * 9

    def fwd():
        for i in range(REPS):
            ops.conv_rows(xs[i % 4], w, b, residual=rs[i % 4], shuffle=r)

    def fwdbwd():
        for i in range(REPS):
            y = ops.conv_rows(xs[i % 4], w, b, residual=rs[i % 4], shuffle=r)
            y.backward(gy)

    with torch.no_grad():
        tf = timed(fwd) / REPS
    tfb = timed(fwdbwd) / REPS
    print(f"conv {cin}->{cout} r={r} {B}x{H}x{W}: fwd {tf*1e3:7.1f} us ({flop/tf/1e9/2.5e3*100:5.1f}% of 2.5 PF)   "
          f"fwd+bwd {tfb*1e3:7.1f} us (bwd {1e3*(tfb-tf):7.1f} us, {2*flop/(tfb-tf)/1e9/2.5e3*100:5.1f}%)", flush=True)
