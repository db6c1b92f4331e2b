"""Debug aid: where does the bf16 backward of the seg-UNet drift from the fp32 one?  Every backward op's output, in order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import segunet_oracle as S
from rdst_amd.loss import seg_unet as U

mode = sys.argv[1] if len(sys.argv) > 1 else "decoder-L1"
layers = [1, 3, 5] if "encoder" in mode else []
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shape = (2, 1, hw, hw)
g = torch.Generator().manual_seed(11)
sr = torch.rand(shape, generator=g)
hr = (sr + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
sd = S.make_unet_weights(1, 4, 0)
traces = {}
for dt in (torch.float32, torch.bfloat16):
    mod = U.SegUNet_F({mode: layers}, "OASIS", allow_random_init=True)
    mod.load_state_dict(sd, strict=True)
    mod.to("cuda:0").set_compute_dtype(dt)
    seq = []
    ob, oc, oa = U._Runner.bn_bwd, U._Runner.conv, U._Runner.bn_apply
    def bnb(self, dy, mask, raw, coef, want_g=False, gadd=None, _o=ob):
        r = _o(self, dy, mask, raw, coef, want_g, gadd)
        seq.append(("bn_bwd C=%d P=%d" % (raw.shape[-1], raw.numel() // raw.shape[-1]), (r[0] if want_g else r).float().cpu()))
        return r
    def cv(self, x1, name, _o=oc, **kw):
        y = _o(self, x1, name, **kw)
        seq.append((("convT " if kw.get("transposed") else "conv  ") + name, y.float().cpu()))
        return y
    def ap(self, x, coef, relu=True, x2=None, coef2=None, res=None, _o=oa):
        y = _o(self, x, coef, relu, x2, coef2, res)
        seq.append(("bn_apply C=%d" % x.shape[-1], y.float().cpu()))
        return y
    U._Runner.bn_bwd, U._Runner.conv, U._Runner.bn_apply = bnb, cv, ap
    s = sr.to("cuda:0").requires_grad_(True)
    l, _ = mod(s, hr.to("cuda:0"))
    l.backward()
    torch.cuda.synchronize()
    U._Runner.bn_bwd, U._Runner.conv, U._Runner.bn_apply = ob, oc, oa
    seq.append(("d sr", s.grad.float().cpu()))
    traces[dt] = (l.item(), seq)
l32, s32 = traces[torch.float32]
l16, s16 = traces[torch.bfloat16]
print("loss", l32, l16)
for (n, a), (n2, b) in zip(s32, s16):
    assert n == n2
    rel = (a - b).norm().item() / max(a.norm().item(), 1e-30)
    cos = (a * b).sum().item() / max(a.norm().item() * b.norm().item(), 1e-30)
    print(f"{n:44s} rel {rel:.3e} cos {cos:.5f} |fp32| {a.norm().item():.3e}")
