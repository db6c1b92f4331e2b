"""fp32 K1 (generic matrix-core kernel, wattn_mfma.hip) at the bench shape: us per launch.
python tools/k1_f32.py [C ...]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W, heads, ws = 32, 64, 64, 6, 8
for C in [int(c) for c in sys.argv[1:]] or [60, 90, 120]:
    q = torch.randn(B, H, W, 3 * C, device=dev)
    o = torch.empty(B, H, W, C, device=dev)
    table = 0.5 * torch.randn(225, heads, device=dev)
    def run():
        _lib.check(lib.rdst_wattn_fwd(q.data_ptr(), 3 * C, table.data_ptr(), None, 0, o.data_ptr(), C, B, H, W, C, heads, ws, 4,
                                      (C // heads) ** -0.5, _lib.F32, torch.cuda.current_stream().cuda_stream), "rdst_wattn_fwd")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C}: {1e3 * e0.elapsed_time(e1) / 5:8.1f} us", flush=True)
