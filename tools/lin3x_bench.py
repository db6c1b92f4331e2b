"""fp32x3 streaming Linear forward (lin3x_mfma.hip) through the C ABI with a PREPACKED image, cold (rotating buffers), us per call
and the fraction of 8 TB/s on its algorithmic bytes (K + N [+ N residual]) * 4 per token.  RDST_HIP_LIB selects an ablation build.
python tools/lin3x_bench.py [old]      (old: workspace = NULL -> the round-5 split kernels of linear_mfma.hip)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0"); M = 131072
old = len(sys.argv) > 1 and sys.argv[1] == "old"
st = torch.cuda.current_stream().cuda_stream
SH = [(60, 180, 1, 0, 0, "qkv60"), (90, 270, 1, 0, 0, "qkv90"), (120, 360, 1, 0, 0, "qkv120"), (60, 60, 0, 0, 1, "proj60"),
      (90, 90, 0, 0, 1, "proj90"), (120, 120, 0, 0, 1, "proj120"), (60, 30, 1, 0, 0, "tail60"), (120, 30, 1, 0, 0, "tail120"),
      (60, 120, 1, 0, 0, "fc1_60"), (90, 180, 1, 0, 0, "fc1_90"), (120, 240, 1, 0, 0, "fc1_120"), (120, 60, 0, 1, 1, "fc2_60"),
      (180, 90, 0, 1, 1, "fc2_90"), (240, 120, 0, 1, 1, "fc2_120")]
tot = 0.0
ONLY = [a for a in sys.argv[1:] if not a.startswith("-") and a != "old"]   # shape names; --once: a single call each (the -DL3X_STAMPS build prints)
ONCE = "--once" in sys.argv
for K, N, ln, act, res, name in SH:
    if ONLY and name not in ONLY:
        continue
    NB = 4
    xs = [torch.randn(M, K, device=dev) for _ in range(NB)]
    ys = [torch.empty(M, N, device=dev) for _ in range(NB)]
    rs = [torch.randn(M, N, device=dev) for _ in range(NB)] if res else None
    w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev) * 0.1
    lw = torch.ones(K, device=dev) if ln else None; lb = torch.zeros(K, device=dev) if ln else None
    stats = torch.empty(M, 2, device=dev)
    nws = lib.rdst_ln_linear_fwd_workspace2(K, N, _lib.F32X3)
    wsp = torch.empty(nws, dtype=torch.uint8, device=dev)

    def call(i, wbytes):
        rc = lib.rdst_ln_linear_fwd(xs[i % NB].data_ptr(), K, lw.data_ptr() if ln else None, lb.data_ptr() if ln else None, act,
                                    w.data_ptr(), b.data_ptr(), rs[i % NB].data_ptr() if res else None, N, ys[i % NB].data_ptr(), N,
                                    stats.data_ptr() if ln else None, None if old else wsp.data_ptr(), 0 if old else wbytes, M, K, N, 1.0,
                                    _lib.F32X3, st)
        assert rc == 0, (rc, lib.rdst_last_error())

    call(0, nws)   # packs
    torch.cuda.synchronize()
    if ONCE:
        print(name, flush=True)
        continue
    for i in range(4):
        call(i, _lib.PREPACKED)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 12
    e0.record()
    for i in range(n):
        call(i, _lib.PREPACKED)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    by = (K + N + (N if res else 0)) * 4 * M
    tot += us
    print(f"{name:9s} {us:7.1f} us   {by / 1e6:6.1f} MB   {by / us / 8e6:5.3f} of 8 TB/s", flush=True)
print(f"sum {tot:.1f} us")
