#!/usr/bin/env python3
"""Forward-only micro-benchmark of the conv op at the cfg2 shapes (fusion conv 150->60, 64x64, B 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd.networks.common import Conv2d
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, H, W = 32, 64, 64
for cin, cout in [(150, 60), (60, 60)]:
    conv = Conv2d(cin, cout, 3, padding=1).to(dev)
    xs = [torch.randn(B, H, W, cin, device=dev).to(dt) for _ in range(4)]
    with torch.no_grad():
        for i in range(12):
            conv.forward_rows(xs[i % 4])
    torch.cuda.synchronize()
    print(cin, cout, flush=True)
