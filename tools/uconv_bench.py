"""Cold timing of rdst_u_conv on the seg-UNet's layer shapes at the bench size (B = 32 of 256x256).
usage: python tools/uconv_bench.py [fp32x3|bf16|fp32] [layer-name-substring]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdst_amd import _lib as L
lib = L.load()
DEV = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32x3"
sub = sys.argv[2] if len(sys.argv) > 2 else ""
code = {"fp32x3": L.F32X3, "bf16": L.BF16, "fp32": L.F32}[mode]
dt = torch.bfloat16 if mode == "bf16" else torch.float32
B = 32
# name, C1, C2, up, Cout, k, stride, Hin (input geometry after upsampling), transposed
LAYERS = [("layer1 64->64", 64, 0, 0, 64, 3, 1, 64, 0), ("layer2 128->128", 128, 0, 0, 128, 3, 1, 32, 0),
          ("layer2.0 64->128 s2", 64, 0, 0, 128, 3, 2, 64, 0), ("layer3 256->256", 256, 0, 0, 256, 3, 1, 16, 0),
          ("layer4 512->512", 512, 0, 0, 512, 3, 1, 8, 0), ("dec0.conv1 768->256", 512, 256, 1, 256, 3, 1, 16, 0),
          ("dec1.conv1 384->128", 256, 128, 1, 128, 3, 1, 32, 0), ("dec2.conv1 192->64", 128, 64, 1, 64, 3, 1, 64, 0),
          ("dec3.conv1 128->32", 64, 64, 1, 32, 3, 1, 128, 0), ("dec3.conv2 32->32", 32, 0, 0, 32, 3, 1, 128, 0),
          ("dec4.conv1 32->16", 32, 0, 1, 16, 3, 1, 256, 0), ("dec4.conv2 16->16", 16, 0, 0, 16, 3, 1, 256, 0),
          ("dec4.conv1 dgrad 16->32", 16, 0, 0, 32, 3, 1, 256, 1), ("dec0.conv1 dgrad 256->768", 256, 0, 0, 768, 3, 1, 16, 1)]
tot = 0.0
for name, c1, c2, up, co, k, s, hin, tr in LAYERS:
    if sub and sub not in name:
        continue
    H = W = hin
    Ho = (H - 1) // s + 1 if not tr else H
    cin = c1 + c2
    npad = (co + 31) // 32 * 32
    NB = 3
    x1 = [torch.randn(B, H // 2 if up else H, W // 2 if up else W, c1, device=DEV).to(dt) for _ in range(NB)]
    x2 = [torch.randn(B, H, W, c2, device=DEV).to(dt) if c2 else None for _ in range(NB)]
    from rdst_amd.loss.seg_unet import _pack_weights
    wp = _pack_weights(torch.randn(k * k, npad, cin), code).to(DEV)
    y = [torch.empty(B, Ho, Ho, co, device=DEV, dtype=dt) for _ in range(NB)]
    def run(i, st):
        L.check(lib.rdst_u_conv(x1[i].data_ptr(), c1, c1, up, None if x2[i] is None else x2[i].data_ptr(), c2, c2, wp.data_ptr(), None, None, 0,
                                y[i].data_ptr(), co, B, H, W, Ho, Ho, co, npad, k, s, tr, code, st, None, None, None), "u_conv")
    run(0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        for i in range(NB):
            run(i, st)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / (5 * NB)
    fl = 2.0 * B * Ho * Ho * co * cin * k * k
    tot += us
    print(f"{name:28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s useful", flush=True)
print("sum", round(tot, 1), "us")
