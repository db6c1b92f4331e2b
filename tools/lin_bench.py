#!/usr/bin/env python3
"""Micro-benchmark of the K3 linear ops at the cfg2 shapes (M = 131072 tokens)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n

dev = torch.device("cuda:0"); M = 131072; dt = torch.bfloat16
for (K, N, ln, act, res, name) in [(60,180,1,0,0,"ln+qkv60"),(120,360,1,0,0,"ln+qkv120"),(60,60,0,0,1,"proj60"),(120,120,0,0,1,"proj120"),
                                   (120,240,1,0,0,"ln+fc1_120"),(240,120,0,1,1,"gelu+fc2_120"),(120,30,1,0,0,"tail120")]:
    x = torch.randn(M, K, device=dev).to(dt).requires_grad_(True)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).requires_grad_(True); b = torch.zeros(N, device=dev, requires_grad=True)
    lw = torch.ones(K, device=dev, requires_grad=True) if ln else None; lb = torch.zeros(K, device=dev, requires_grad=True) if ln else None
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    gy = torch.randn(M, N, device=dev).to(dt)
    with torch.no_grad():
        tf = t(lambda: ops.ln_linear(x, lw, lb, w, b, in_act=act, residual=r))
    y = ops.ln_linear(x, lw, lb, w, b, in_act=act, residual=r)
    tb = t(lambda: torch.autograd.grad(y, [x, w, b] + ([lw, lb] if ln else []), gy, retain_graph=True))
    by_f = M * (K + N + (N if res else 0)) * 2
    print(f"{name:14s} fwd {tf:7.1f} us {by_f/tf/1e3:7.1f} GB/s | bwd(all) {tb:7.1f} us", flush=True)
