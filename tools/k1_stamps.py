"""One eager K1 launch at C for the in-kernel stamps of the debug library:
RDST_HIP_LIB=$PWD/rdst_amd/librdst_hip_dbg.so RDST_K1_STAMPS=1 [RDST_K1_DEBUG=7] python tools/k1_stamps.py 60"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import ops
dev = torch.device("cuda:0")
for C in [int(c) for c in sys.argv[1:]] or [60]:
    q = torch.randn(32, 64, 64, 3 * C, device=dev).bfloat16()
    table = 0.5 * torch.randn(225, 6, device=dev)
    with torch.no_grad():
        ops.window_attention(q, table, 64, 64, 6, 8, 4, (C // 6) ** -0.5)
    torch.cuda.synchronize()
