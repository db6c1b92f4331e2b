#!/usr/bin/env python3
"""One fused-Mlp backward per width at M = 131072 (for in-kernel stamps: RDST_HIP_LIB=..._dbg.so RDST_MLP_STAMPS=3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import _lib
lib = _lib.load()
dev = "cuda:0"
M = 131072
for C in (60, 90, 120):
    hid = 2 * C
    x = torch.randn(M, C, device=dev).bfloat16()
    gy = torch.randn(M, C, device=dev).bfloat16()
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = torch.randn(hid, C, device=dev) * C ** -0.5, torch.zeros(hid, device=dev)
    w2 = torch.randn(C, hid, device=dev) * hid ** -0.5
    xf = x.float()
    stats = torch.stack([xf.mean(-1), (xf.var(-1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
    dx = torch.empty_like(x)
    G = [torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty(C, device=dev), torch.empty_like(lw), torch.empty_like(lb)]
    nb = lib.rdst_mlp_bwd_workspace(M, C, hid)
    wsp = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(lib.rdst_mlp_bwd(x.data_ptr(), C, lw.data_ptr(), lb.data_ptr(), stats.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                    w2.data_ptr(), gy.data_ptr(), C, dx.data_ptr(), C, G[0].data_ptr(), G[1].data_ptr(), G[2].data_ptr(),
                                    G[3].data_ptr(), G[4].data_ptr(), G[5].data_ptr(), wsp.data_ptr(), nb, M, C, hid, _lib.BF16, st), "mlp_bwd")
    torch.cuda.synchronize()
    print("C", C, flush=True)
