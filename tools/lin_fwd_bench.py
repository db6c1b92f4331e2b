#!/usr/bin/env python3
"""Forward-only micro-benchmark of the fused LN/act -> Linear -> residual op at the cfg2 shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops
dev = torch.device("cuda:0"); M = 131072; dt = torch.bfloat16
for (K, N, ln, act, res, name) in [(60,180,1,0,0,"ln+qkv60"),(120,360,1,0,0,"ln+qkv120"),(60,60,0,0,1,"proj60"),(120,120,0,0,1,"proj120"),
                                   (120,240,1,0,0,"ln+fc1_120"),(240,120,0,1,1,"gelu+fc2_120")]:
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(6)]
    w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.zeros(N, device=dev)
    lw = torch.ones(K, device=dev) if ln else None; lb = torch.zeros(K, device=dev) if ln else None
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    with torch.no_grad():
        for i in range(18):
            ops.ln_linear(xs[i % 6], lw, lb, w, b, in_act=act, residual=r)
    torch.cuda.synchronize()
    print(name, flush=True)
