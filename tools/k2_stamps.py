"""One eager K2 launch per width for the in-kernel stamps of the debug library:
RDST_HIP_LIB=$PWD/rdst_amd/librdst_hip_dbg.so RDST_K2_STAMPS=1 python tools/k2_stamps.py 120"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W, heads, ws = 32, 64, 64, 6, 8
for C in [int(c) for c in sys.argv[1:]] or [60, 90, 120]:
    q = torch.randn(B, H, W, 3 * C, device=dev).bfloat16()
    g = torch.randn(B, H, W, C, device=dev).bfloat16()
    d = torch.empty_like(q)
    table = 0.5 * torch.randn(225, heads, device=dev)
    dtab = torch.zeros(225, heads, device=dev)
    nws = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
    wsp = torch.empty(nws, dtype=torch.uint8, device=dev)
    _lib.check(lib.rdst_wattn_bwd(q.data_ptr(), 3 * C, table.data_ptr(), None, 0, g.data_ptr(), C, d.data_ptr(), 3 * C,
                                  dtab.data_ptr(), wsp.data_ptr(), nws, B, H, W, C, heads, ws, 4, (C // heads) ** -0.5,
                                  _lib.BF16, torch.cuda.current_stream().cuda_stream), "rdst_wattn_bwd")
    torch.cuda.synchronize()
    print("C", C, flush=True)
