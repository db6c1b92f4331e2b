"""fp32x3 one-pass Linear backward (lnlin3x_mfma.hip) through the C ABI, cold (rotating buffers): us per call incl. its slab sums, and the
fraction of 8 TB/s on its algorithmic bytes (x K + dY N + dX K [+ dX_add K]) * 4 per token.  RDST_HIP_LIB selects another build
(the -DRDST_DEBUG library honours RDST_LBX_OFF=1: the round-5 three-launch path).   python tools/lnlin3x_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdst_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0"); M = 131072
st = torch.cuda.current_stream().cuda_stream
SH = [(60, 180, 1, 0, 1, "qkv60"), (90, 270, 1, 0, 1, "qkv90"), (120, 360, 1, 0, 1, "qkv120"), (60, 60, 0, 0, 0, "proj60"),
      (90, 90, 0, 0, 0, "proj90"), (120, 120, 0, 0, 0, "proj120"), (60, 30, 1, 0, 0, "tail60"), (120, 30, 1, 0, 0, "tail120"),
      (60, 120, 1, 0, 1, "fc1_60"), (90, 180, 1, 0, 1, "fc1_90"), (120, 240, 1, 0, 1, "fc1_120"), (120, 60, 0, 1, 0, "fc2_60"),
      (180, 90, 0, 1, 0, "fc2_90"), (240, 120, 0, 1, 0, "fc2_120")]
tot = 0.0
BF = "--bf16" in sys.argv     # the bf16 kernels (lnlin3_mfma.hip) at their shapes, for comparison
if BF:
    SH = [t for t in SH if t[2] == 1 and t[1] in (3 * t[0], 30) or t[2] == 0 and t[3] == 0]
DT, CODE, EB = (torch.bfloat16, _lib.BF16, 2) if BF else (torch.float32, _lib.F32X3, 4)
ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]   # shape names; --once: a single call each (the -DLBX_STAMPS build prints)
ONCE = "--once" in sys.argv
for K, N, ln, act, add, name in SH:
    if ONLY and name not in ONLY:
        continue
    NB = 3
    xs = [torch.randn(M, K, device=dev).to(DT) for _ in range(NB)]
    dys = [torch.randn(M, N, device=dev).to(DT) for _ in range(NB)]
    dxs = [torch.empty(M, K, device=dev, dtype=DT) for _ in range(NB)]
    adds = [torch.randn(M, K, device=dev).to(DT) for _ in range(NB)] if add else None
    w = torch.randn(N, K, device=dev) * K ** -0.5
    lw = torch.ones(K, device=dev) if ln else None; lb = torch.zeros(K, device=dev) if ln else None
    stats = torch.stack([xs[0].float().mean(1), torch.rsqrt(xs[0].float().var(1, unbiased=False) + 1e-5)], 1).contiguous() if ln else None
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    dlw = torch.empty(K, device=dev) if ln else None; dlb = torch.empty(K, device=dev) if ln else None
    nws = lib.rdst_ln_linear_bwd_workspace(M, K, N)
    wsp = torch.empty(nws, dtype=torch.uint8, device=dev)
    P = lambda t: t.data_ptr() if t is not None else None

    def call(i):
        rc = lib.rdst_ln_linear_bwd(xs[i % NB].data_ptr(), K, P(lw), P(lb), P(stats), act, w.data_ptr(), dys[i % NB].data_ptr(), N,
                                    dxs[i % NB].data_ptr(), K, adds[i % NB].data_ptr() if add else None, K, dw.data_ptr(), db.data_ptr(),
                                    P(dlw), P(dlb), wsp.data_ptr(), nws, M, K, N, 1.0, CODE, st)
        assert rc == 0, (rc, lib.rdst_last_error())

    if ONCE:
        print(name, flush=True)
        call(0); torch.cuda.synchronize()
        continue
    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 9
    e0.record()
    for i in range(n):
        call(i)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    by = (2 * K + N + (K if add else 0)) * EB * M
    tot += us
    print(f"{name:9s} {us:7.1f} us   {by / 1e6:6.1f} MB   {by / us / 8e6:5.3f} of 8 TB/s", flush=True)
print(f"sum {tot:.1f} us")
