#!/usr/bin/env python3
"""Kernel times of the one-output-channel tail conv (60 -> 1, 3x3, 256x256, B = 32, bf16) under rocprofv3:
bash tools/prof_tool.sh c1 tools/conv_c1_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, H, W, Cin = 32, 256, 256, 60
xs = [torch.randn(B, H, W, Cin, device=dev).to(dt).requires_grad_(True) for _ in range(3)]
w = (torch.randn(1, Cin, 3, 3, device=dev) * (Cin * 9) ** -0.5).requires_grad_(True)
b = torch.zeros(1, device=dev, requires_grad=True)
gy = torch.randn(B, H, W, 1, device=dev).to(dt)
for it in range(4):
    for x in xs:
        y = ops.conv_rows(x, w, b)
        y.backward(gy)
torch.cuda.synchronize()
print("ok")
