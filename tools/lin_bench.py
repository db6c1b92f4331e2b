#!/usr/bin/env python3
"""Forward micro-benchmark of the fused LayerNorm/Linear op at the E1 shapes (M = 131072 tokens)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
M = 131072
for K in (60, 90, 120):
    for kind, N in (("qkv", 3 * K), ("proj", K), ("tail", 30)):
        ln = kind != "proj"
        xs = [torch.randn(32, 4096, K, device=dev).to(dt) for _ in range(4)]
        rs = [torch.randn(32, 4096, N, device=dev).to(dt) for _ in range(4)] if kind == "proj" else [None] * 4
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b = torch.zeros(N, device=dev)
        lw, lb = (torch.ones(K, device=dev), torch.zeros(K, device=dev)) if ln else (None, None)
        with torch.no_grad():
            for i in range(12):
                ops.ln_linear(xs[i % 4], lw, lb, w, b, residual=rs[i % 4])
        torch.cuda.synchronize()
        print(kind, K, N, flush=True)
