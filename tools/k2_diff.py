"""K2 through the C ABI against the oracle's autograd (CPU fp32), per q/k/v section and head.
usage: python tools/k2_diff.py [C] [B H W shift]"""
import sys, torch
sys.path.insert(0, ".")
from rdst_amd import _lib
from oracle import rdst_oracle as O
lib = _lib.load()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B, H, W, shift = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (1, 24, 16, 4)
heads, ws = 6, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
q = torch.randn(B, H, W, 3 * C).bfloat16()
g = torch.randn(B, H, W, C).bfloat16()
table = 0.5 * torch.randn(225, heads)
scale = (C // heads) ** -0.5
qr = q.float().requires_grad_(True)
tr = table.clone().requires_grad_(True)
O.window_attention_core(qr, tr, heads, ws, shift, scale).backward(g.float())
qd, gd, td = q.to(dev), g.to(dev), table.to(dev)
d = torch.zeros(B, H, W, 3 * C, device=dev, dtype=torch.bfloat16)
dt = torch.zeros(225, heads, device=dev)
nws = lib.rdst_wattn_bwd_workspace(B, H, W, C, heads, ws)
wsp = torch.zeros(nws, dtype=torch.uint8, device=dev)
rc = lib.rdst_wattn_bwd(qd.data_ptr(), 3 * C, td.data_ptr(), None, 0, gd.data_ptr(), C, d.data_ptr(), 3 * C, dt.data_ptr(),
                        wsp.data_ptr(), nws, B, H, W, C, heads, ws, shift, scale, _lib.BF16, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert rc == 0, rc
a, b = d.float().cpu(), qr.grad
D = C // heads
for si, nm in enumerate("qkv"):
    for hd in range(heads):
        x = a[..., si * C + hd * D: si * C + (hd + 1) * D]
        y = b[..., si * C + hd * D: si * C + (hd + 1) * D]
        bad = ~torch.isfinite(x)
        err = ((x - y).norm() / y.norm()).item() if not bad.any() else float("nan")
        print(f"d{nm} head {hd}: rel {err:.3e} nonfinite {int(bad.sum())}", end="")
        if bad.any():
            print("  first", bad.nonzero()[:4].tolist(), end="")
        print()
print("dtable rel", ((dt.cpu() - tr.grad).norm() / tr.grad.norm()).item())
if len(sys.argv) > 6 or True:
    x = a[0, :8, :8, :D] - b[0, :8, :8, :D]          # dq head 0, window 0 (shift 0 only meaningful)
    y = b[0, :8, :8, :D]
    print("dq h0 per-channel rel err:", [round(float(x[..., c].norm() / y[..., c].norm()), 3) for c in range(D)])
    print("dq h0 per-token rel err (8x8):")
    e = (x.norm(dim=-1) / y.norm(dim=-1))
    for r in range(8):
        print("  ", [round(float(v), 2) for v in e[r]])
