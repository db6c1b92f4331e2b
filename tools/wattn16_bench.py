"""Times the window-16 attention forward / backward through the C ABI (B x 128 x 128 tokens, C = 60/90/120).
usage: python tools/wattn16_bench.py [B] [bf16|fp32]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
for C in (60, 90, 120):
    for shift in (0, 8):
        qkv = torch.randn(B, 128, 128, 3 * C, device=dev).to(DT).requires_grad_(True)
        table = (0.5 * torch.randn(961, 6, device=dev)).requires_grad_(True)
        go = torch.randn(B, 128, 128, C, device=dev).to(DT)
        for it in range(2):
            o = ops.window_attention(qkv, table, 128, 128, 6, 16, shift, (C // 6) ** -0.5)
            o.backward(go)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 5
        tf = tb = 0.0
        for it in range(n):
            e[0].record()
            o = ops.window_attention(qkv, table, 128, 128, 6, 16, shift, (C // 6) ** -0.5)
            e[1].record()
            o.backward(go)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        print(f"C={C} shift={shift}: fwd {tf / n * 1e3:8.1f} us   bwd {tb / n * 1e3:8.1f} us", flush=True)
