import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops
dev = torch.device("cuda:0"); M = 131072; dt = torch.float32 if os.environ.get("LIN_F32") else torch.bfloat16
K, N = int(sys.argv[1]), int(sys.argv[2]); ln = int(sys.argv[3])
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0; res = int(sys.argv[5]) if len(sys.argv) > 5 else 0
x = torch.randn(M, K, device=dev).to(dt)
w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.zeros(N, device=dev)
lw = torch.ones(K, device=dev) if ln else None; lb = torch.zeros(K, device=dev) if ln else None
rr = torch.randn(M, N, device=dev).to(dt) if res else None
with torch.no_grad():
    for i in range(6):
        ops.ln_linear(x, lw, lb, w, b, in_act=act, residual=rr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(6):
        ops.ln_linear(x, lw, lb, w, b, in_act=act, residual=rr)
    e1.record()
torch.cuda.synchronize()
print(f"K={K} N={N} ln={ln} act={act} res={res}: {1e3 * e0.elapsed_time(e1) / 6:7.1f} us per call", flush=True)
