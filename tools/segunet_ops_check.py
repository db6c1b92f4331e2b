"""Debug aid: strict relative-L2 check of the rdst_u_* elementwise ops vs torch (fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from rdst_amd import _lib as L
lib = L.load()
DEV = "cuda:0"
st = lambda: torch.cuda.current_stream().cuda_stream
def nhwc(x): return x.permute(0, 2, 3, 1).contiguous().to(DEV)
def nchw(r): return r.float().cpu().permute(0, 3, 1, 2)
def rel(a, b): return (a - b).norm().item() / max(b.norm().item(), 1e-30)
g = torch.Generator().manual_seed(0)
scratch = torch.empty(lib.rdst_u_scratch_bytes(), dtype=torch.uint8, device=DEV)
for C, (B, H, W) in [(16, (2, 64, 64)), (64, (2, 16, 16)), (512, (2, 2, 2)), (32, (2, 32, 32))]:
    x = 1.5 * torch.randn(B, C, H, W, generator=g) + 0.3
    bn = torch.nn.BatchNorm2d(C).train()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g)); bn.bias.copy_(0.1 * torch.randn(C, generator=g))
    xr = x.clone().requires_grad_(True)
    y = F.relu(bn(xr))
    dy = torch.randn(B, C, H, W, generator=g) * 1e-5
    y.backward(dy)
    P = B * H * W
    dx_, ddy = nhwc(x), nhwc(dy)
    coef = torch.empty(4 * C, device=DEV)
    gam, bet = bn.weight.detach().to(DEV), bn.bias.detach().to(DEV)
    L.check(lib.rdst_u_bn_stats(dx_.data_ptr(), C, P, C, gam.data_ptr(), bet.data_ptr(), 1e-5, 0.1, None, None,
                                coef.data_ptr(), scratch.data_ptr(), 0, st()), "s")
    out = torch.empty(B, H, W, C, device=DEV)
    L.check(lib.rdst_u_bn_apply(dx_.data_ptr(), C, coef.data_ptr(), None, 0, None, None, 0, 1, out.data_ptr(), C, P, C, 0, st()), "a")
    dxo = torch.empty(B, H, W, C, device=DEV)
    L.check(lib.rdst_u_bn_bwd(ddy.data_ptr(), C, out.data_ptr(), C, dx_.data_ptr(), C, coef.data_ptr(), dxo.data_ptr(), C, None, 0, None, 0, P, C,
                              scratch.data_ptr(), 0, st()), "b")
    torch.cuda.synchronize()
    mean = x.mean((0, 2, 3)); var = x.var((0, 2, 3), unbiased=False)
    cc = coef.cpu()
    print(f"C={C} P={P}: y rel {rel(nchw(out), y.detach()):.2e}  dx rel {rel(nchw(dxo), xr.grad):.2e}  mean err {(cc[2*C:3*C]-mean).abs().max():.2e} rstd rel {((cc[3*C:]-(var+1e-5).rsqrt())/(var+1e-5).rsqrt()).abs().max():.2e}")
# pair loss
for C, P in [(16, 8192), (64, 512)]:
    a = torch.randn(P, C, generator=g).relu(); b = torch.randn(P, C, generator=g).relu()
    for mse in (0, 1):
        ar = a.clone().requires_grad_(True)
        l = F.mse_loss(ar, b) if mse else F.l1_loss(ar, b)
        l.backward()
        da, db_ = a.to(DEV), b.to(DEV)
        loss = torch.zeros((), device=DEV)
        L.check(lib.rdst_u_pair_loss_fwd(da.data_ptr(), C, db_.data_ptr(), C, P, C, mse, 1.0, 0, loss.data_ptr(), scratch.data_ptr(), 0, st()), "p")
        ga = torch.empty(P, C, device=DEV)
        L.check(lib.rdst_u_pair_loss_bwd(da.data_ptr(), C, db_.data_ptr(), C, P, C, mse, 1.0, None, None, 0, ga.data_ptr(), C, 0, st()), "pb")
        torch.cuda.synchronize()
        print(f"pair C={C} mse={mse}: loss {loss.item():.7f} vs {l.item():.7f}  grad rel {rel(ga.cpu(), ar.grad):.2e}")
