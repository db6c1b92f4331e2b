"""fp32 parity mode, fused LN/act -> Linear -> residual forward and backward at the E1 shapes: us per call
(torch events around 6 calls on rotating inputs; forward, then forward + backward).
python tools/lin_f32.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import ops
dev = torch.device("cuda:0"); M = 131072; dt = torch.float32
ops.set_f32_split(len(sys.argv) > 1 and sys.argv[1] == "x3")   # python tools/lin_f32.py x3: the split-bf16 arithmetic


def timed(fn, n=6):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for (K, N, ln, act, res, name) in [(60, 180, 1, 0, 0, "ln+qkv60"), (90, 270, 1, 0, 0, "ln+qkv90"), (120, 360, 1, 0, 0, "ln+qkv120"),
                                   (60, 60, 0, 0, 1, "proj60"), (90, 90, 0, 0, 1, "proj90"), (120, 120, 0, 0, 1, "proj120"),
                                   (120, 30, 1, 0, 0, "tail120"), (60, 120, 1, 0, 0, "ln+fc1_60"), (90, 180, 1, 0, 0, "ln+fc1_90"),
                                   (120, 240, 1, 0, 0, "ln+fc1_120"), (120, 60, 0, 1, 1, "gelu+fc2_60"),
                                   (180, 90, 0, 1, 1, "gelu+fc2_90"), (240, 120, 0, 1, 1, "gelu+fc2_120")]:
    xs = [torch.randn(M, K, device=dev, dtype=dt, requires_grad=True) for _ in range(3)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).requires_grad_(True); b = torch.zeros(N, device=dev, requires_grad=True)
    lw = torch.ones(K, device=dev, requires_grad=True) if ln else None
    lb = torch.zeros(K, device=dev, requires_grad=True) if ln else None
    r = torch.randn(M, N, device=dev, dtype=dt) if res else None
    gy = torch.randn(M, N, device=dev, dtype=dt)

    def fwd(i):
        with torch.no_grad():
            ops.ln_linear(xs[i % 3], lw, lb, w, b, in_act=act, residual=r)

    def both(i):
        ops.ln_linear(xs[i % 3], lw, lb, w, b, in_act=act, residual=r).backward(gy)

    tf, tb = timed(fwd), timed(both)
    fl = 2.0 * M * K * N
    print(f"{name:14s} fwd {tf:7.1f} us ({fl / tf / 157e6:4.2f} of 157 TF)   fwd+bwd {tb:7.1f} us ({3 * fl / tb / 157e6:4.2f})", flush=True)
