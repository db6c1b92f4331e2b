"""fp32 K2 (generic matrix-core kernel, wattn_bwd_mfma.hip) at the bench shape: us per launch.
RDST_HIP_LIB=...dbg.so RDST_K2_DEBUG=<bits> python tools/k2_f32.py [x3] [C ...]   (x3: the split-bf16 arithmetic)"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W, heads, ws = 32, 64, 64, 6, 8
CODE = _lib.F32X3 if "x3" in sys.argv[1:] else _lib.F32
for C in [int(c) for c in sys.argv[1:] if c != "x3"] or [60, 120]:
    q = torch.randn(B, H, W, 3 * C, device=dev)
    g = torch.randn(B, H, W, C, device=dev)
    d = torch.empty_like(q)
    table = 0.5 * torch.randn(225, heads, device=dev)
    dtab = torch.zeros(225, heads, device=dev)
    nws = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
    wsp = torch.empty(nws, dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.rdst_wattn_bwd(q.data_ptr(), 3 * C, table.data_ptr(), None, 0, g.data_ptr(), C, d.data_ptr(), 3 * C,
                                      dtab.data_ptr(), wsp.data_ptr(), nws, B, H, W, C, heads, ws, 4, (C // heads) ** -0.5,
                                      CODE, torch.cuda.current_stream().cuda_stream), "rdst_wattn_bwd")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C}: {1e3 * e0.elapsed_time(e1) / 5:8.1f} us", flush=True)
