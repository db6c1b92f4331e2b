#!/usr/bin/env python3
"""Micro-benchmark of the window-attention kernels (K1 forward, K2 backward) at the cfg2 shapes:
B=32, 64x64, ws 8, heads 6, C in {60, 90, 120}, bf16 or fp32.  Prints per-launch time from HIP events on
the launch stream and the achieved algorithmic GB/s (4*C*elt fwd, 7*C*elt bwd per token)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bwd", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cs", default="60,90,120")
    ap.add_argument("--rot", type=int, default=0, help="fwd only: rotate over this many qkv buffers (cold HBM reads), "
                    "all launches back to back inside ONE event pair")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda:0")
    B, H, W, heads, ws = a.batch, 64, 64, 6, 8
    for C in [int(c) for c in a.cs.split(",")]:
        for shift in (0, 4):
            if a.rot:
                qs = [torch.randn(B, H, W, 3 * C, device=dev).to(dt) for _ in range(a.rot)]
                table = 0.5 * torch.randn(225, heads, device=dev)
                scale = (C // heads) ** -0.5
                with torch.no_grad():
                    for q in qs:
                        ops.window_attention(q, table, H, W, heads, ws, shift, scale)
                    torch.cuda.synchronize()
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(a.iters * a.rot):
                        ops.window_attention(qs[i % a.rot], table, H, W, heads, ws, shift, scale)
                    e1.record()
                    torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / (a.iters * a.rot)
                nb = B * H * W * 4 * C * qs[0].element_size()
                print(f"C={C:3d} shift={shift} {a.dtype} fwd cold x{a.rot} {us:8.1f} us {nb / us / 1e3:8.1f} GB/s", flush=True)
                del qs
                continue
            qkv = torch.randn(B, H, W, 3 * C, device=dev).to(dt).requires_grad_(True)
            table = (0.5 * torch.randn(225, heads, device=dev)).requires_grad_(True)
            go = torch.randn(B, H, W, C, device=dev).to(dt)
            scale = (C // heads) ** -0.5
            for _ in range(3):
                o = ops.window_attention(qkv, table, H, W, heads, ws, shift, scale)
                if a.bwd:
                    o.backward(go)
            kt = ops.KernelTimer()
            ops.set_kernel_timer(kt)
            for _ in range(a.iters):
                o = ops.window_attention(qkv, table, H, W, heads, ws, shift, scale)
                if a.bwd:
                    o.backward(go)
            torch.cuda.synchronize()
            ops.set_kernel_timer(None)
            f = kt.summary("fwd")
            line = f"C={C:3d} shift={shift} {a.dtype} fwd {1e3 * f['total_ms'] / f['launches']:8.1f} us " \
                   f"{f['bytes'] / f['total_ms'] / 1e6:8.1f} GB/s"
            if a.bwd:
                b = kt.summary("bwd")
                line += f" | bwd {1e3 * b['total_ms'] / b['launches']:8.1f} us {b['bytes'] / b['total_ms'] / 1e6:8.1f} GB/s"
            print(line, flush=True)


if __name__ == "__main__":
    main()
