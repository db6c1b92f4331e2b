"""One-shot driver of the fused Mlp backward (stamps / timing): python tools/mlp_one.py C [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0"); M = 131072
C = int(sys.argv[1]); hid = 2 * C; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
x = torch.randn(M, C, device=dev).bfloat16(); dy = torch.randn(M, C, device=dev).bfloat16()
lw = torch.ones(C, device=dev); lb = torch.zeros(C, device=dev)
w1 = torch.randn(hid, C, device=dev) * C ** -0.5; b1 = torch.zeros(hid, device=dev)
w2 = torch.randn(C, hid, device=dev) * hid ** -0.5
xf = x.float(); stats = torch.stack([xf.mean(-1), (xf.var(-1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
dx = torch.empty_like(x)
G = [torch.empty_like(t) for t in (w1, b1, w2, lb, lw, lb)]
nb = lib.rdst_mlp_bwd_workspace(M, C, hid); wsp = torch.empty(nb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.rdst_mlp_bwd(x.data_ptr(), C, lw.data_ptr(), lb.data_ptr(), stats.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                w2.data_ptr(), dy.data_ptr(), C, dx.data_ptr(), C, *[g.data_ptr() for g in G], wsp.data_ptr(), nb,
                                M, C, hid, _lib.BF16, st), "mlp_bwd")
b2 = torch.zeros(C, device=dev); y = torch.empty_like(x); st2 = torch.empty(M, 2, device=dev)
def runf():
    _lib.check(lib.rdst_mlp_fwd(x.data_ptr(), C, lw.data_ptr(), lb.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                b2.data_ptr(), y.data_ptr(), C, st2.data_ptr(), M, C, hid, _lib.BF16, st), "mlp_fwd")
if len(sys.argv) > 3 and sys.argv[3] == "fwd":
    run = runf
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
print(f"C={C}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per call (kernel + sum + finish)")
