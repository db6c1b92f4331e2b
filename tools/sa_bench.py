"""K8 (fused attention half of a Swin block) against the three launches it replaces, full size (32 x 64 x 64), HIP events.
usage: python tools/sa_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from rdst_amd import _lib
from test_swinattn_gpu import _params, _fused, _unfused, DEV
from util import rand

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, W = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 64, 64
M = B * H * W
for C in (60, 90, 120):
    scale = (C // 6) ** -0.5
    P = {k: v.to(DEV).contiguous() for k, v in _params(C, 3).items()}
    xs = [rand((M, C), 7 + i).to(DEV).bfloat16() for i in range(4)]
    for shift in (0, 4):
        out = {}
        bufs = [tuple(torch.empty((M, n), dtype=dt, device=DEV) for n, dt in ((3 * C, torch.bfloat16), (C, torch.bfloat16), (C, torch.bfloat16), (2, torch.float32)))
                for _ in range(4)]
        for name, fn in (("fused", _fused), ("three", _unfused)):
            # one call that packs the weight images into workspaces that the timed calls then use as PREPACKED
            if name == "fused":
                wsp = torch.empty(lib.rdst_swin_attn_fwd_workspace(C), dtype=torch.uint8, device=DEV)
                lib.rdst_swin_attn_fwd(xs[0].data_ptr(), C, P["n1w"].data_ptr(), P["n1b"].data_ptr(), P["qkvw"].data_ptr(), P["qkvb"].data_ptr(),
                                       P["table"].data_ptr(), P["projw"].data_ptr(), P["projb"].data_ptr(), bufs[0][0].data_ptr(), 3 * C,
                                       bufs[0][1].data_ptr(), C, bufs[0][2].data_ptr(), C, bufs[0][3].data_ptr(), wsp.data_ptr(), wsp.numel(),
                                       B, H, W, C, 6, 8, shift, scale, _lib.BF16, torch.cuda.current_stream().cuda_stream)
            else:
                w1 = torch.empty(lib.rdst_ln_linear_fwd_workspace(C, 3 * C), dtype=torch.uint8, device=DEV)
                w2 = torch.empty(lib.rdst_ln_linear_fwd_workspace(C, C), dtype=torch.uint8, device=DEV)
                st = torch.cuda.current_stream().cuda_stream
                lib.rdst_ln_linear_fwd(xs[0].data_ptr(), C, P["n1w"].data_ptr(), P["n1b"].data_ptr(), 0, P["qkvw"].data_ptr(), P["qkvb"].data_ptr(),
                                       None, 0, bufs[0][0].data_ptr(), 3 * C, bufs[0][3].data_ptr(), w1.data_ptr(), w1.numel(), M, C, 3 * C, 1.0, _lib.BF16, st)
                lib.rdst_ln_linear_fwd(bufs[0][1].data_ptr(), C, None, None, 0, P["projw"].data_ptr(), P["projb"].data_ptr(), xs[0].data_ptr(), C,
                                       bufs[0][2].data_ptr(), C, None, w2.data_ptr(), w2.numel(), M, C, C, 1.0, _lib.BF16, st)
                wsp = (w1, w2)
            for i in range(3):
                fn(lib, _lib, xs[i % 4], C, P, B, H, W, C, shift, scale, out=bufs[i % 4], wsp=wsp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                fn(lib, _lib, xs[i % 4], C, P, B, H, W, C, shift, scale, out=bufs[i % 4], wsp=wsp)
            e1.record()
            torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) / reps * 1e3
        # (weights prepacked, outputs preallocated: 4 buffer sets in rotation)
        print(f"C={C} shift={shift}: fused {out['fused']:.1f} us   three calls {out['three']:.1f} us   bytes fused {6*C*2*M/1e6:.1f} MB -> "
              f"{6*C*2*M/out['fused']/1e6:.2f} TB/s", flush=True)
