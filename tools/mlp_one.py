#!/usr/bin/env python3
"""One fused-Mlp forward per width at M = 131072 (for stamps / profiles)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import _lib
lib = _lib.load()
dev = "cuda:0"
M = 131072
for C in (60, 90, 120):
    hid = 2 * C
    x = torch.randn(M, C, device=dev).bfloat16()
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = torch.randn(hid, C, device=dev) * C ** -0.5, torch.zeros(hid, device=dev)
    w2, b2 = torch.randn(C, hid, device=dev) * hid ** -0.5, torch.zeros(C, device=dev)
    y = torch.empty_like(x); stats = torch.empty(M, 2, device=dev)
    nws = lib.rdst_mlp_fwd_workspace(C, hid); wsp = torch.empty(nws, dtype=torch.uint8, device=dev)
    for _ in range(3):
        _lib.check(lib.rdst_mlp_fwd(x.data_ptr(), C, lw.data_ptr(), lb.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                    y.data_ptr(), C, stats.data_ptr(), wsp.data_ptr(), nws, M, C, hid, _lib.BF16,
                                    torch.cuda.current_stream().cuda_stream), "mlp_fwd")
    torch.cuda.synchronize()
    print("C", C, flush=True)
