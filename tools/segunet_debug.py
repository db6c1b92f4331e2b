"""Debug aid: gradient w.r.t. every UNet stage, HIP path vs the torch oracle (fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import segunet_oracle as S
from rdst_amd.loss import seg_unet as U

mode = sys.argv[1] if len(sys.argv) > 1 else "decoder"
shape = (2, 1, 64, 64)
g = torch.Generator().manual_seed(11)
sr = torch.rand(shape, generator=g)
hr = (sr + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
sd = S.make_unet_weights(1, 4, 0)
mod = U.SegUNet_F({mode: []}, "OASIS", allow_random_init=True)
mod.load_state_dict(sd, strict=True)
mod.to("cuda:0")

# oracle with retained intermediate grads
srr = sr.clone().requires_grad_(True)
bn = S.BNState(sd, False)
feats = S.encoder_forward(srr, sd, bn)
for f in feats[1:]:
    f.retain_grad()
fs = list(feats[1:])[::-1]
x, skips = fs[0], fs[1:]
outs = []
for i in range(5):
    p = f"decoder.blocks.{i}"
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    if i < len(skips):
        x = torch.cat([x, skips[i]], dim=1)
    r1 = F.conv2d(x, sd[p + ".conv1.0.weight"], None, 1, 1); r1.retain_grad()
    a1 = F.relu(bn(r1, p + ".conv1.1")); a1.retain_grad()
    r2 = F.conv2d(a1, sd[p + ".conv2.0.weight"], None, 1, 1); r2.retain_grad()
    x = F.relu(bn(r2, p + ".conv2.1"))
    x.retain_grad()
    outs.append(x)
    if i == 4:
        inter = (r1, a1, r2, x)
with torch.no_grad():
    hdec = S.unet_forward(hr, sd, "decoder", S.BNState(sd, False))
loss = F.l1_loss(outs[-1], hdec)
loss.backward()

# HIP path with a tapped runner
taps = {}
orig_bwd = U._Runner.backward
orig_sum = None
def tapped(self, save, d_feats, d_dec=None, upstream=None):
    lib = self.lib
    real = lib.rdst_u_sumpool2
    cnt = [0]
    def sp(*a):
        rc = real(*a)
        taps[f"dprev{cnt[0]}"] = (a[4], a[6:10])   # dX ptr, B,H,W,C
        cnt[0] += 1
        return rc
    lib.rdst_u_sumpool2 = sp
    realmp = lib.rdst_u_maxpool_bwd
    try:
        return orig_bwd(self, save, d_feats, d_dec, upstream)
    finally:
        lib.rdst_u_sumpool2 = real
U._Runner.backward = tapped
keep = []
orig_conv = U._Runner.conv
def conv_tap(self, x1, name, **kw):
    y = orig_conv(self, x1, name, **kw)
    if kw.get("transposed"):
        taps.setdefault("convT:" + name, y)
    return y
U._Runner.conv = conv_tap
seq = []
orig_bnb = U._Runner.bn_bwd
def bnb_tap(self, dy, mask, raw, coef, want_g=False, gadd=None):
    r = orig_bnb(self, dy, mask, raw, coef, want_g, gadd)
    seq.append(("bn", dy, r[0] if want_g else r, mask, raw, coef))
    return r
U._Runner.bn_bwd = bnb_tap
srg = sr.to("cuda:0").requires_grad_(True)
got, _ = mod(srg, hr.to("cuda:0"))
got.backward()
torch.cuda.synchronize()
print("loss", got.item(), loss.item())
def rel(a, b):
    return (a - b).norm().item() / max(b.norm().item(), 1e-20)
print("d sr rel", rel(srg.grad.cpu(), srr.grad))
r1, a1, r2, xo = inter
c = lambda t: t.float().cpu().permute(0, 3, 1, 2)
print("block4 d_dec  ", rel(c(seq[0][1]), xo.grad))
print("block4 dr2    ", rel(c(seq[0][2]), r2.grad))
_, dy_, dr_, mk_, raw_, cf_ = seq[0]
dy64, mk64, raw64 = dy_.double().cpu().reshape(-1, 16), mk_.double().cpu().reshape(-1, 16), raw_.double().cpu().reshape(-1, 16)
mean = raw64.mean(0); var = raw64.var(0, unbiased=False); rstd = (var + 1e-5).rsqrt()
gam = sd["decoder.blocks.4.conv2.1.weight"].double()
g64 = dy64 * (mk64 > 0)
xh = (raw64 - mean) * rstd
ref64 = gam * rstd * (g64 - g64.mean(0) - xh * (g64 * xh).mean(0))
print("manual fp64 BN bwd from HIP inputs: HIP rel", rel(dr_.double().cpu().reshape(-1, 16), ref64), " oracle rel", rel(r2.grad.permute(0, 2, 3, 1).reshape(-1, 16).double(), ref64))
o64 = r2.grad.permute(0, 2, 3, 1).reshape(-1, 16).double()
h64 = dr_.double().cpu().reshape(-1, 16)
print("per-channel rel", ((h64 - o64).norm(dim=0) / o64.norm(dim=0)))
ro = r2.detach().permute(0, 2, 3, 1).reshape(-1, 16).double()
print("per-channel raw rel", ((raw64 - ro).norm(dim=0) / ro.norm(dim=0)))
xo64 = xo.detach().permute(0, 2, 3, 1).reshape(-1, 16).double()
go = xo.grad.permute(0, 2, 3, 1).reshape(-1, 16).double() * (xo64 > 0)
print("sum g hip/oracle", g64.sum(0), go.sum(0))
mo = ro.mean(0); vo = ro.var(0, unbiased=False); xho = (ro - mo) * (vo + 1e-5).rsqrt()
print("sum gxh hip/oracle", (g64 * xh).sum(0), (go * xho).sum(0))
cf = cf_.cpu().double()
print("coef a err", (cf[:16] - gam * rstd).abs().max().item(), "mean err", (cf[32:48] - mean).abs().max().item(), "rstd relerr", ((cf[48:64] - rstd) / rstd).abs().max().item())
print("raw vs oracle r2", rel(raw_.float().cpu().permute(0, 3, 1, 2), r2.detach()), " mask mismatch", ((mk_.float().cpu().permute(0, 3, 1, 2) > 0) != (xo.detach() > 0)).sum().item())
print("block4 da1    ", rel(c(seq[1][1]), a1.grad))
print("block4 dr1    ", rel(c(seq[1][2]), r1.grad))
# gradients w.r.t. inputs of the first conv of each stage = transposed conv outputs
for i in range(5):
    name = f"decoder.blocks.{i}.conv1.0"
    dcat = taps["convT:" + name].float().cpu().permute(0, 3, 1, 2)
    cx = [512, 256, 128, 64, 32][i]
    if i < 4:
        ref_skip = feats[4 - i].grad   # includes the encoder-path gradient too: only comparable for the decoder-only part
    up = outs[i - 1].grad if i > 0 else feats[5].grad
    got_prev = F.avg_pool2d(dcat[:, :cx], 2) * 4
    print(f"block {i}: d(prev out) rel {rel(got_prev, up):.3e}")
for k in range(1, 6):
    pass
for li, nb in zip((4, 3, 2, 1), (3, 6, 4, 3)):
    name = f"encoder.layer{li}.0.conv1"
    y = taps["convT:" + name].float().cpu().permute(0, 3, 1, 2)
    ref = feats[li].grad if li >= 2 else None
    if ref is not None:
        print(f"layer{li} input grad (features[{li}]) rel {rel(y, ref):.3e}")
