"""One-shot driver of the fused LayerNorm-Linear backward: python tools/lnlin_one.py K N [ln]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops
dev = torch.device("cuda:0"); M = 131072
K, N = int(sys.argv[1]), int(sys.argv[2]); ln = int(sys.argv[3]) if len(sys.argv) > 3 else 1
x = torch.randn(1, M, K, device=dev).bfloat16().requires_grad_(True)
w = (torch.randn(N, K, device=dev) * K ** -0.5).requires_grad_(True); b = torch.zeros(N, device=dev, requires_grad=True)
lw = torch.ones(K, device=dev, requires_grad=True) if ln else None; lb = torch.zeros(K, device=dev, requires_grad=True) if ln else None
gy = torch.randn(1, M, N, device=dev).bfloat16()
for i in range(6):
    y = ops.ln_linear(x, lw, lb, w, b)
    y.backward(gy)
torch.cuda.synchronize()
