#!/usr/bin/env python3
"""bench.py — RDST-E1 x4 training-step throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic LR patches on every rank:
forward (RDSTSR, HIP kernels) + L1 loss + backward + flat-bucket gradient all-reduce (RCCL, N > 1)
+ Adam.  Workload = BASELINE.json configs[1]: RDST-E1 x4 (config_files/RDST_E1_OASIS_example_SRx4.ini
:188-240), batch 32 per GPU of 1x64x64 patches -> 256x256, bf16 activations.  Rank 0 prints ONE JSON
line.  Weak scaling: per-GPU batch fixed.

`--config` selects the workload: e1 (default, the BASELINE.json metric), ws16 (BASELINE configs[3]), e1_unetf (the ini's
'UNet-F' fine-tuning state: 0.1 L1 + 1 SegUNet_F({'encoder-L1': [1]})), e1_hrl (BASELINE configs[4], RDST-HRL: 0.1 L1 +
1 SegUNet_F({'label-hr': []}): resnet34-UNet + multiclass Dice in the backward path).  `--backend gloo` runs the N > 1
path without RCCL (tests; both ranks may share one GPU with RDST_BENCH_ONE_GPU=1).

Extra objects on the line (N = 1, rank 0):
  roofline     — the window-attention forward kernel (K1): algorithmic bytes (4*C*elt per token per launch,
                 DESIGN.md) / its average duration, vs 8 TB/s HBM.  `frac` is the COLD figure: the step's 48 K1
                 launches (recorded through the C ABI while the step graph was captured) replayed back to back on
                 the step's own buffers, HIP events on the launch stream.  `in_step` = the same launches inside the
                 forward pass, by difference of two HIP graphs (with / without the K1 calls): qkv was just written.
                 `traffic` = HBM bytes per launch from the PMC counters (profiles/pmc_latest.json, collected by
                 tools/pmc_collect.py on THIS kernel source, else null).
  roofline_bwd — K2 the same way (cold).
  kernels      — the six most expensive C-ABI ops of the step: launches, average us, algorithmic bytes / FLOPs,
                 fraction of the bounding peak, all measured live by the same replay.
  fp32_mode    — throughput of the fp32 parity mode (the mode that carries the 4-decimal PSNR claim).
  parity_mode  — the same step in the fp32x3 mode (fp32 tensors, split-bf16 GEMMs): the fast mode that holds the 4-decimal bar.
  cpu_baseline — the CPU oracle (oracle/rdst_oracle.py, a port) timed on the host cores on a bounded
                 sample (batch 4) of the same workload.
  configs      — (default run only: --config e1, N = 1) the OTHER BASELINE.json configurations on the same line, 5 timed
                 graph steps each after capture: ws16 (configs[3]: value, ms_per_step, K1 / K2 roofline fractions),
                 e1_unetf and e1_hrl (configs[4]: value, ms_per_step, the loss network's share of the step), each with its
                 own cpu_baseline (oracle + oracle/segunet_oracle.py at batch 1).  `--no-configs` skips them.
On N > 1 the line also carries world_size_seen, backend, bucket_bytes and the measured all-reduce time (HIP events around 5
calls of FlatGradBucket.all_reduce_mean) so that a scaling record explains itself.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

E1 = dict(img_size=64, patch_size=1, in_chans=1, sr_scale=4, embed_dim=60, dense_layer_depths=[2] * 8,
          num_heads=[6] * 8, window_size=[8] * 8, rdb_depths=[3] * 8, mlp_ratio=2., qkv_bias=True, qk_scale=None,
          growth_rate=30, dense_scale=1., dim_modify_mode='tail', rdb_residual_scale=1., global_res_scale=1.,
          resi_connection='1conv', pre_norm=True, feature_last_operation=True)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3,   # dense peaks, MI355X_MICROARCH.md "Chip-level parameters"
                    "fp32x3": 1250.0}                # two bf16 MFMAs per fp32 product


# ---------------------------------------------------------------------------------------------------------------------
# Measurement aid (bench-side only; nothing in rdst_amd/ knows about it): every call through the C ABI is recorded while
# the training step is captured into its HIP graph.  The graph's private memory pool keeps every operand alive at a fixed
# address, so the very same launches can be replayed later, op by op, back to back inside one HIP-event pair.
# ---------------------------------------------------------------------------------------------------------------------
def _alg(name, a, elt):
    """(algorithmic HBM bytes, FLOPs) of one C-ABI call from its arguments — the per-unit figures of DESIGN.md section 4
    times the units of the launch.  `a` = the positional arguments as ops.py passed them."""
    if name == "rdst_wattn_fwd":        # read qkv (3C) + write out (C) per token
        B, H, W, C = a[7:11]
        return B * H * W * 4 * C * elt, 4 * B * H * W * a[12] * a[12] * C
    if name == "rdst_swin_attn_fwd":    # K8: read x (C); write qkv (3C), a (C), x1 (C) per token (+ 8 bytes of LayerNorm statistics)
        B, H, W, C = a[18:22]
        M, N = B * H * W, a[23] * a[23]
        return M * (6 * C * elt + 8), 2 * M * C * 3 * C + 4 * M * N * C + 2 * M * C * C
    if name == "rdst_wattn_fwd_lse":    # window 16 with the row statistics kept: + 4 bytes per (token, head)
        B, H, W, C = a[6:10]
        return B * H * W * (4 * C * elt + 4 * a[10]), 4 * B * H * W * a[11] * a[11] * C
    if name == "rdst_wattn_bwd_lse":    # reads qkv (3C) + dout (C) + the forward's output (C) + statistics, writes dqkv (3C)
        B, H, W, C = a[13:17]
        return B * H * W * (8 * C * elt + 4 * a[17]), 10 * B * H * W * a[18] * a[18] * C
    if name == "rdst_wattn_bwd":        # read qkv (3C) + dout (C), write dqkv (3C)
        B, H, W, C = a[12:16]
        return B * H * W * 7 * C * elt, 10 * B * H * W * a[17] * a[17] * C
    if name == "rdst_ln_linear_fwd":
        M, K, N = a[14:17]
        return M * (K + N + (N if a[7] else 0)) * elt, (2 * M * K * N if a[5] else 0)
    if name in ("rdst_ln_linear_bwd", "rdst_ln_linear_bwd2"):   # (bwd2: + the second addend's K channels per token)
        M, K, N = a[19:22]
        add2 = K if (name.endswith("2") and a[25]) else 0
        return (M * (K + N + (K if a[9] else 0) + (K if a[11] else 0) + add2) * elt,
                2 * M * K * N * ((1 if a[9] else 0) + (1 if a[13] else 0)) if a[6] else 0)
    if name == "rdst_mlp_fwd":
        M, C, hid = a[13:16]
        return 2 * M * C * elt, 4 * M * C * hid
    if name == "rdst_mlp_bwd":
        M, C, hid = a[20:23]
        return 3 * M * C * elt, 10 * M * C * hid
    if name == "rdst_conv_fwd":
        B, H, W, Cin, Cout, k = a[11:17]
        P = B * H * W
        return P * (Cin + Cout + (Cout if a[5] else 0)) * elt, 2 * P * Cin * Cout * k * k
    if name == "rdst_conv_bwd":
        B, H, W, Cin, Cout, k = a[14:20]
        P = B * H * W
        return (P * (Cin + Cout + (Cin if a[6] else 0)) * elt,
                2 * P * Cin * Cout * k * k * ((1 if a[6] else 0) + (1 if a[10] else 0)))
    return 0, 0


class Recorder:
    """Wraps the C-ABI entry points of the loaded library: records (name, args) of every call and can leave the calls of
    one entry point out (`skip`), which is how an op's time INSIDE the step is measured (step with - step without)."""
    OPS = ("rdst_swin_attn_fwd", "rdst_wattn_fwd", "rdst_wattn_bwd", "rdst_wattn_fwd_lse", "rdst_wattn_bwd_lse", "rdst_ln_linear_fwd", "rdst_ln_linear_bwd", "rdst_ln_linear_bwd2", "rdst_mlp_fwd",
           "rdst_mlp_bwd", "rdst_conv_fwd", "rdst_conv_bwd", "rdst_nchw_to_rows", "rdst_rows_to_nchw")
    STREAM_ARG = {"rdst_ln_linear_bwd2": 24}   # position of the stream argument where it is not the last one

    def __init__(self, lib):
        self.lib, self.calls, self.skip, self.orig = lib, [], None, {}

    def __enter__(self):
        for n in self.OPS:
            f = getattr(self.lib, n)
            self.orig[n] = f

            def wrap(*a, _n=n, _f=f):
                self.calls.append((_n, a))
                return 0 if _op_key(_n) == self.skip else _f(*a)
            setattr(self.lib, n, wrap)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.lib, n, f)
        return False


def _timed_replay(graph, n):
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _replay_calls(lib, calls, reps=3):
    """Average duration (ms) of the recorded launches replayed back to back from one HIP graph (cold operands: a step's
    activations are several GB, far beyond the 256 MiB Infinity Cache)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        st = torch.cuda.current_stream().cuda_stream
        with torch.cuda.graph(g, stream=side):
            st = torch.cuda.current_stream().cuda_stream
            for n, a in calls:                      # same operands, the capturing stream
                i = Recorder.STREAM_ARG.get(n, len(a) - 1)
                getattr(lib, n)(*a[:i], st, *a[i + 1:])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return _timed_replay(g, reps) / len(calls)


def _source_hash(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "rdst_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_traffic(kernel_key, files):
    """HBM bytes per launch from the committed PMC collection (profiles/pmc_latest.json, written by tools/pmc_collect.py:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) — only
    if it was collected on the kernel sources this run was built from; else null with the reason."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        ent = d["kernels"][kernel_key]
        if ent.get("source_hash") != _source_hash(files):
            return None, f"profiles/pmc_latest.json was collected on other sources of {files[0]} (hash {ent.get('source_hash')})"
        return int(ent["bytes_per_launch"]), f"profiles/pmc_latest.json ({ent.get('how', '')})"
    except Exception as e:  # noqa: BLE001
        return None, f"no PMC collection ({type(e).__name__})"


# BASELINE.json configs[3]: RDST x2, 3-channel (BraTS-style), 128x128 -> 256x256, window 16 (`--config ws16`; a parity /
# stress configuration, not the headline workload: its window attention runs on the shape-generic kernels)
WS16 = dict(E1, img_size=128, in_chans=3, sr_scale=2, window_size=[16] * 8)
CONFIGS = {"e1": (E1, 64, 1, 4, "RDST-E1 x4 (RDST_E1_OASIS_example_SRx4.ini), 1x64x64 LR patches -> 256x256"),
           "ws16": (WS16, 128, 3, 2, "RDST x2 3-channel, window 16 (BASELINE configs[3]), 3x128x128 LR patches -> 256x256"),
           "e1_unetf": (E1, 64, 1, 4, "RDST-E1 x4, 'UNet-F' state of RDST_E1_OASIS_example_SRx4.ini:34,46 (0.1 L1 + 1 SegUNet_F "
                                      "encoder-L1 [1]), 1x64x64 LR patches -> 256x256"),
           "e1_hrl": (E1, 64, 1, 4, "RDST-HRL x4 (BASELINE configs[4]): 0.1 L1 + 1 SegUNet_F label-hr (resnet34-UNet + multiclass "
                                    "Dice in the backward path), 1x64x64 LR patches -> 256x256")}
LOSS_MODES = {"e1_unetf": {"encoder-L1": [1]}, "e1_hrl": {"label-hr": []}}


def build_loss(config, device):
    """None (plain L1) or the reference's weighted SRLoss for the UNet-F states (loss/sr_loss.py:35-51, ini :31-46).  The
    UNet checkpoint (loss/unet_oasis.pt) is not in the repository: seeded random initialisation, said so in `data`."""
    if config not in LOSS_MODES:
        return None
    import types
    from rdst_amd.loss.sr_loss import SRLoss
    torch.manual_seed(1)
    paras = types.SimpleNamespace(gpu_id=device.index, precision=False, training_losses=["L1", "UNet-F"],
                                  loss_scalars={"UNet-F": {"L1": 0.1, "UNet-F": 1}}, training_states=["UNet-F"],
                                  unet_loss_layers=LOSS_MODES[config], unet_loss_mode="OASIS", unet_allow_random=True)
    return SRLoss(paras)


def build_net(device, dtype, cfg=None):
    from rdst_amd.networks.rdst_variations import RDSTSR
    torch.manual_seed(0)                       # seeded default init, as SURVEY.md §8d prescribes
    net = RDSTSR(**(cfg or E1))
    net.to(device).train().set_compute_dtype(dtype)
    return net


def _cpu_baseline_worker(threads, sample_batch, timed, which="e1"):
    """Runs in a child process: the oracle (a CPU port of the reference algorithm), fwd + L1 + bwd."""
    torch.set_num_threads(threads)
    from oracle import rdst_oracle as O
    cfg, cin, lr, sr = {"e1": (O.CFG_E1, 1, 64, 4), "tiny": (O.CFG_TINY, 1, 64, 4), "e1_unetf": (O.CFG_E1, 1, 64, 4),
                        "e1_hrl": (O.CFG_E1, 1, 64, 4),
                        "ws16": (O.make_cfg(**{**O.CFG_WS16, "img_size": 128}), 3, 128, 2)}[which]
    unet = None
    if which in LOSS_MODES:     # 0.1 L1 + 1 SegUNet_F (oracle/segunet_oracle.py: the restated smp resnet34-UNet, seeded weights)
        from oracle import segunet_oracle as SO
        (mode, layers), = LOSS_MODES[which].items()
        unet = (SO, {k: v.clone() for k, v in SO.make_unet_weights(1, 4, seed=1).items()}, mode, layers)
    sd = O.make_weights(cfg, 0)
    lay = O.state_dict_layout(cfg)
    sd = {k: (v.clone().requires_grad_(True) if lay[k][2] not in ("index", "mask", "shift_w", "shift_b") else v)
          for k, v in sd.items()}
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(sample_batch, cin, lr, lr, generator=g)
    tgt = torch.rand(sample_batch, cin, lr * sr, lr * sr, generator=g)
    best = None
    for it in range(1 + timed):
        t0 = time.perf_counter()
        y = O.rdstsr_forward(x, sd, cfg)
        loss = F.l1_loss(y, tgt)
        if unet is not None:
            SO, usd, mode, layers = unet
            loss = 0.1 * loss + SO.segunet_loss(y, tgt, usd, mode, layers)
        loss.backward()
        dt = time.perf_counter() - t0
        if it > 0:
            best = dt if best is None else min(best, dt)
    print(json.dumps({"best_s": best}), flush=True)


def _host_cpu():
    """(model name, physical cores, logical CPUs) of the host from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None and core is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1)), (logical or (os.cpu_count() or 1))


def cpu_baseline(which="e1", threads=16, sample_batch=4, timed=3, timeout_s=150, start_only=False):
    """Bounded CPU baseline next to the GPU number.  16 threads: on the 256-CPU GPU-box host the
    oracle is FASTEST there (measured 1.7 s/step at 16 threads, 2.9 s at 32, 5.7 s at 64: the ops are
    small and OpenMP fork/join dominates beyond that), so this is the best CPU figure, not a handicap."""
    import subprocess
    threads = max(1, min(threads, os.cpu_count() or 1))
    model, phys, logical = _host_cpu()
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), str(sample_batch), str(timed), which]
    names = {"e1": "RDST-E1 x4, 1x64x64", "tiny": "RDST-E tiny x4 (BASELINE configs[0]), 1x64x64", "ws16": "RDST x2 window 16, 3x128x128",
             "e1_unetf": "RDST-E1 x4 + 0.1 L1 + SegUNet_F encoder-L1 [1] (oracle/segunet_oracle.py), 1x64x64",
             "e1_hrl": "RDST-E1 x4 + 0.1 L1 + SegUNet_F label-hr (oracle/segunet_oracle.py: resnet34-UNet x 2 forward + backward + Dice), 1x64x64"}
    host = {"cores": threads, "host_cpu": model, "host_physical_cores": phys, "host_logical_cpus": logical}
    if start_only:    # the caller collects later (cpu_baseline_collect): the CPU sample runs while the GPU is being measured
        return (subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT), host, names[which], threads,
                sample_batch, timed, timeout_s, time.time())
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        best = json.loads(r.stdout.strip().splitlines()[-1])["best_s"]
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "patches/s", **host, "kind": "port",
                "sample": f"failed or exceeded {timeout_s}s: {type(e).__name__}"}
    return {"value": round(sample_batch / best, 3), "unit": "patches/s", **host, "kind": "port",
            "sample": f"oracle/rdst_oracle.py fwd+L1+bwd, {names[which]}, batch {sample_batch} fp32, "
                      f"best of {timed} after 1 warm-up, torch CPU, {threads} threads (the oracle's fastest thread count on this host)"}


def cpu_baseline_collect(h):
    """Result of a cpu_baseline(..., start_only=True) child."""
    proc, host, name, threads, sample_batch, timed, timeout_s, t0 = h
    try:
        so, _ = proc.communicate(timeout=max(1.0, timeout_s - (time.time() - t0)))
        best = json.loads(so.strip().splitlines()[-1])["best_s"]
    except Exception as e:  # noqa: BLE001
        try:
            proc.kill()
        except Exception:  # noqa: BLE001
            pass
        return {"value": None, "unit": "patches/s", **host, "kind": "port", "sample": f"failed or exceeded {timeout_s}s: {type(e).__name__}"}
    return {"value": round(sample_batch / best, 3), "unit": "patches/s", **host, "kind": "port",
            "sample": f"oracle fwd+loss+bwd, {name}, batch {sample_batch} fp32, best of {timed} after 1 warm-up, torch CPU, {threads} threads, "
                      f"the three configs[3] / configs[4] samples timed side by side (16 threads each of {host['host_logical_cpus']} CPUs)"}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-baseline-worker":
        _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "e1")
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="patches per GPU (default 32; 8 for --config ws16)")
    ap.add_argument("--config", default="e1", choices=sorted(CONFIGS), help="e1 = the BASELINE.json metric (default)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp32x3"],
                    help="bf16 = the BASELINE.json mode; fp32 = exact parity mode; fp32x3 = fp32 tensors, split-bf16 GEMMs (the fast parity mode)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--unet-dtype", default=None, choices=["fp32", "fp32x3", "bf16"],
                    help="arithmetic of the seg-UNet loss network (e1_unetf / e1_hrl); default fp32x3: fp32 activations, 3-term bf16 "
                         "split on the matrix cores (the features of SR and HR are DIFFERENCED: bf16 activations are too noisy)")
    ap.add_argument("--force-pg", action="store_true",
                    help="initialise the process group and run its collectives (broadcast, bucket all-reduce) even with ONE rank: "
                         "executes the RCCL path on a one-GPU box (launch through torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=3, help="replay passes over the recorded launches")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-op measurements (clean per-step kernel profiles)")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the fp32 parity-mode measurement (field fp32_mode)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (field configs) of the default run")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # one rank per GPU: the launch and the flag must agree
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node "
              f"{args.gpus} (or run plain `python bench.py` for one GPU)", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("RDST_BENCH_ONE_GPU") == "1":      # tests: every rank on cuda:0 (gloo only)
        if args.backend != "gloo":
            print("bench.py: RDST_BENCH_ONE_GPU=1 needs --backend gloo", file=sys.stderr)
            sys.exit(2)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_pg = world > 1 or args.force_pg
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    from rdst_amd import _lib, dp, optim  # noqa: F401
    if args.force_pg:
        dp.FORCE_COLLECTIVES = True
    from rdst_amd.trainer import DPTrainStep
    dtype = torch.bfloat16 if args.dtype == "bf16" else ("fp32x3" if args.dtype == "fp32x3" else torch.float32)
    cfg, lr_size, in_ch, sr, cfg_name = CONFIGS[args.config]
    if args.batch is None:
        args.batch = 8 if args.config == "ws16" else 32
    net = build_net(device, dtype, cfg)
    loss_obj = build_loss(args.config, device)
    unet_dtype = None
    if loss_obj is not None:
        unet_dtype = args.unet_dtype or "fp32x3"
        loss_obj.loss_functions["UNet-F"].set_compute_dtype(unet_dtype)
    # the trainer-step shell a user calls (rdst_amd/trainer.py): flat bucket, one all-reduce, fused Adam with
    # utils/optim.py:30-53's hyper-parameters from the ini; forward + loss + backward replayed from ONE HIP graph
    tr = DPTrainStep(net, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, loss_fn=loss_obj, graph=False)
    bucket = tr.bucket
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    x = torch.rand(B, in_ch, lr_size, lr_size, generator=g).to(device)
    tgt = torch.rand(B, in_ch, lr_size * sr, lr_size * sr, generator=g).to(device)

    lib = _lib.load()
    rec = Recorder(lib)
    # the other BASELINE configurations ride on the default run's line
    want_configs = (rank == 0 and world == 1 and args.config == "e1" and args.dtype == "bf16" and not args.no_configs
                    and not args.no_roofline)
    if args.graph:
        for _ in range(2):          # eager steps: every lazy initialisation (weight-image plans, workspaces) happens here
            tr.step(x, tgt)
        torch.cuda.synchronize()
        tr.use_graph = True
        tr.capture_hook = lambda: rec   # the step graph's pool keeps every recorded operand alive
        if not tr.capture(x, tgt) and rank == 0:
            print("bench.py: graph capture failed; running eagerly", file=sys.stderr)
        tr.capture_hook = None
    recorded = rec.calls if tr.graph is not None else []
    x, tgt = (tr._static if tr.graph is not None else (x, tgt))   # no per-step input copy: the step reads the static buffers

    def sync():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(x, tgt)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(x, tgt)
    sync()
    elapsed = time.perf_counter() - t0
    loss_t = tr._loss_buf.detach().clone().double()
    param_sync = None
    if use_pg:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        dist.all_reduce(loss_t, op=dist.ReduceOp.SUM)
        loss_t /= world
        # replicas must hold identical parameters after the timed steps (same all-reduced gradients, same Adam)
        cs = tr.optimizer.flat_param.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        param_sync = bool(lo.item() == hi.item())
    loss_val = float(loss_t.item())
    comm = None
    if use_pg:   # what the collective costs on this fabric: HIP events around 5 all-reduces of the flat bucket
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bucket.all_reduce_mean(tr.group)
        sync()
        e0.record()
        for _ in range(5):
            bucket.all_reduce_mean(tr.group)
        e1.record()
        sync()
        tms = torch.tensor([e0.elapsed_time(e1) / 5], device=device, dtype=torch.float64)
        dist.all_reduce(tms, op=dist.ReduceOp.MAX)
        comm = {"world_size_seen": dist.get_world_size(), "backend": dist.get_backend(), "bucket_bytes": bucket.nbytes,
                "all_reduce_ms": round(tms.item(), 4),
                "all_reduce_how": "HIP events around 5 x FlatGradBucket.all_reduce_mean (one collective over the flat fp32 bucket), max over ranks"}

    metric = {"e1": "SR patches/sec fwd+bwd, RDST-E1 x4 64->256",
              "ws16": "SR patches/sec fwd+bwd, RDST x2 window-16 128->256 (BASELINE configs[3])",
              "e1_unetf": "SR patches/sec fwd+bwd, RDST-E1 x4 64->256, UNet-F state (seg-UNet stem loss in the backward path)",
              "e1_hrl": "SR patches/sec fwd+bwd, RDST-HRL x4 64->256 (BASELINE configs[4]: seg-UNet label-hr Dice loss)"}[args.config]
    step_desc = "L1" if loss_obj is None else "0.1 L1 + 1 UNet-F"
    out = {
        "metric": metric,
        "value": round(world * B * args.steps / elapsed, 3),
        "unit": "patches/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic" + ("" if loss_obj is None else " (seg-UNet: seeded random init, the reference's loss/unet_oasis.pt is not in the repository)"),
        "config": {"workload": cfg_name + f", step = fwd + {step_desc} + bwd + flat-bucket grad all-reduce + Adam",
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                   "hip_graph": tr.graph is not None, "grad_bucket_bytes": bucket.nbytes,
                   "backend": (args.backend if use_pg else None), "unet_dtype": unet_dtype},
        "loss": round(loss_val, 6),
    }
    if param_sync is not None:
        out["param_sync"] = param_sync
    if comm is not None:
        out.update(comm)
    if loss_obj is not None and "UNet-F" in loss_obj.loss_functions:
        # the frozen loss network is replicated: its train-mode BatchNorm statistics are per rank, and
        # every rank must have run it the same number of times
        nb = loss_obj.loss_functions["UNet-F"].encoder.bn1.num_batches_tracked.detach().to(device).double().reshape(1)
        lo, hi = nb.clone(), nb.clone()
        if use_pg:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out["unet_bn_batches_tracked"] = [int(lo.item()), int(hi.item())]

    if rank == 0 and world == 1 and not args.no_roofline and tr.graph is not None and recorded:
        elt = 2 if args.dtype == "bf16" else 4
        reps = max(1, args.roofline_steps)
        out.update(op_tables(lib, recorded, elt, reps, args.dtype, net, x, ws16=(args.config == "ws16")))
    if rank == 0 and world == 1 and not args.no_roofline:
        if args.dtype == "bf16" and not args.no_fp32_line and args.config == "e1":
            # the parity mode (fp32 activations, exact-fp32 MFMA): the only mode with the 4-decimal PSNR claim
            del tr
            tr = None
            try:
                out["fp32_mode"] = fp32_line(device, x, tgt, B, lib)
            except Exception as e:  # noqa: BLE001
                out["fp32_mode"] = {"value": None, "note": f"failed: {type(e).__name__}: {e}"}
            try:   # the fast parity mode: fp32 tensors, split-bf16 GEMMs (tests/test_fp32x3_gpu.py holds its |dPSNR| < 5e-5 assert)
                out["parity_mode"] = fp32_line(device, x, tgt, B, lib, split=True)
            except Exception as e:  # noqa: BLE001
                out["parity_mode"] = {"value": None, "note": f"failed: {type(e).__name__}: {e}"}
        if want_configs:
            tr = net = bucket = loss_obj = None
            rec.calls = recorded = []
            torch.cuda.empty_cache()
            out["configs"] = {}
            for name in ("ws16", "e1_unetf", "e1_hrl"):
                try:
                    line = extra_config(name, device, lib, out["ms_per_step"])
                except Exception as e:  # noqa: BLE001
                    line = {"value": None, "note": f"failed: {type(e).__name__}: {e}"}
                out["configs"][name] = line
        if not args.no_cpu_baseline:
            if args.config == "ws16":
                out["cpu_baseline"] = cpu_baseline("ws16", sample_batch=1, timed=1, timeout_s=200)
            else:
                out["cpu_baseline"] = cpu_baseline("e1")
                if args.config == "e1":
                    c1 = cpu_baseline("tiny", timeout_s=60)
                    out["cpu_baseline"]["cfg1"] = {"value": c1["value"], "unit": "patches/s", "sample": c1["sample"]}
            if want_configs and "configs" in out:
                # the three samples run side by side (16 threads each), after every GPU measurement of this run is over: a CPU
                # child next to a timed GPU region slowed the headline step by 25 % (measured, round 4)
                hs = {n: cpu_baseline(n, sample_batch=1, timed=1, timeout_s=240, start_only=True) for n in out["configs"]}
                for n, h in hs.items():
                    out["configs"][n]["cpu_baseline"] = cpu_baseline_collect(h)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


def extra_config(name, device, lib, e1_ms, steps=5):
    """One of the other BASELINE configurations, measured the way the headline is (DPTrainStep, forward + loss + backward
    replayed from one HIP graph, flat-bucket Adam), `steps` timed steps after 2 eager steps + capture + 1 replay."""
    from rdst_amd.trainer import DPTrainStep
    cfg, lr_size, in_ch, sr, cfg_name = CONFIGS[name]
    B = 8 if name == "ws16" else 32
    net = build_net(device, torch.bfloat16, cfg)
    loss_obj = build_loss(name, device)
    if loss_obj is not None:
        loss_obj.loss_functions["UNet-F"].set_compute_dtype("fp32x3")
    tr = DPTrainStep(net, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, loss_fn=loss_obj, graph=False)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(B, in_ch, lr_size, lr_size, generator=g).to(device)
    tgt = torch.rand(B, in_ch, lr_size * sr, lr_size * sr, generator=g).to(device)
    for _ in range(2):
        tr.step(x, tgt)
    torch.cuda.synchronize()
    rec = Recorder(lib)
    tr.use_graph = True
    tr.capture_hook = lambda: rec
    ok = tr.capture(x, tgt)
    tr.capture_hook = None
    xs, ts = tr._static if ok else (x, tgt)
    tr.step(xs, ts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(xs, ts)
    torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / steps
    line = {"value": round(B / d, 3), "unit": "patches/s", "ms_per_step": round(1e3 * d, 3), "steps": steps, "per_gpu_batch": B,
            "hip_graph": bool(ok), "dtype": "bf16", "loss": round(float(tr._loss_buf.item()), 6), "workload": cfg_name}
    if loss_obj is not None:
        line["unet_dtype"] = "fp32x3"
        line["loss_network_ms"] = round(1e3 * d - e1_ms, 3)
        line["loss_network_how"] = "this step minus the headline step of the same run (same network, same batch, L1 only)"
        line["data"] = "synthetic (seg-UNet: seeded random init, the reference's loss/unet_oasis.pt is not in the repository)"
    if name == "ws16" and ok and rec.calls:
        for key, op in (("roofline", "rdst_wattn_fwd"), ("roofline_bwd", "rdst_wattn_bwd")):
            calls = [(n, a) for n, a in rec.calls if _op_key(n) == op]
            ms = _replay_calls(lib, calls, 2)
            nbytes = sum(_alg(n, a, 2)[0] for n, a in calls) / len(calls)
            ach = nbytes / (ms * 1e-3) / 1e9
            tr_b, tr_src = _pmc_traffic("wattn16_fwd_kernel" if key == "roofline" else "wattn16_bwd1_kernel", ["wattn16_mfma.hip", "wattn_hd.h"])
            line[key] = {"kernel": op + " (window 16)", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_us": round(1e3 * ms, 2), "launches": len(calls),
                         "traffic": tr_b, "traffic_source": tr_src}
    del tr, net, rec
    torch.cuda.empty_cache()
    if name == "ws16":   # the same configuration in the reference's own arithmetic (exact-fp32 matrix-core kernels, wattn16_f32.hip)
        try:
            line["fp32_mode"] = _fp32_step_line(device, cfg, x, tgt, B, steps=3)
        except Exception as e:  # noqa: BLE001
            line["fp32_mode"] = {"value": None, "note": f"failed: {type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return line


def _fp32_step_line(device, cfg, x, tgt, B, steps):
    from rdst_amd.trainer import DPTrainStep
    net32 = build_net(device, torch.float32, cfg)
    t32 = DPTrainStep(net32, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, graph=False)
    for _ in range(2):
        t32.step(x, tgt)
    torch.cuda.synchronize()
    t32.use_graph = True
    ok = t32.capture(x, tgt)
    xs, ts = t32._static if ok else (x, tgt)
    t32.step(xs, ts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t32.step(xs, ts)
    torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / steps
    return {"value": round(B / d, 3), "unit": "patches/s", "ms_per_step": round(1e3 * d, 3), "steps": steps, "per_gpu_batch": B,
            "hip_graph": bool(ok), "dtype": "fp32", "loss": round(float(t32._loss_buf.item()), 6)}


def fp32_line(device, x, tgt, B, lib, split=False):
    """The same step in the fp32 parity mode (split: the fp32x3 mode), graph-captured like the bf16 one, with its own per-op table."""
    from rdst_amd import ops
    from rdst_amd.trainer import DPTrainStep
    net32 = build_net(device, "fp32x3" if split else torch.float32)
    return _fp32_line(device, x, tgt, B, lib, split, net32)


def _fp32_line(device, x, tgt, B, lib, split, net32):
    from rdst_amd.trainer import DPTrainStep
    t32 = DPTrainStep(net32, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, graph=False)
    for _ in range(2):
        t32.step(x, tgt)
    torch.cuda.synchronize()
    rec = Recorder(lib)
    t32.use_graph = True
    t32.capture_hook = lambda: rec
    t32.capture(x, tgt)
    t32.capture_hook = None
    xs, ts = t32._static if t32.graph is not None else (x, tgt)
    t32.step(xs, ts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        t32.step(xs, ts)
    torch.cuda.synchronize()
    d32 = (time.perf_counter() - t0) / n
    line = {"value": round(B / d32, 3), "unit": "patches/s", "ms_per_step": round(1e3 * d32, 3), "steps": n,
            "hip_graph": t32.graph is not None, "dtype": "fp32x3" if split else "fp32",
            "note": ("fp32 activations, every GEMM operand as two bf16 terms (16 mantissa bits) on v_mfma_f32_32x32x16_bf16, fp32 "
                     "accumulation, softmax / LayerNorm / GELU in fp32: |dPSNR| vs the oracle 3e-6 dB at this shape "
                     "(tests/test_fp32x3_gpu.py asserts < 5e-5), same step and graph capture as the bf16 line") if split else
                    "fp32 activations + exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the parity mode, same step and graph capture as the bf16 line"}
    if t32.graph is not None and rec.calls:
        # (split: two bf16 MFMAs per fp32 product -> half the dense bf16 peak is this mode's matrix-core ceiling)
        tab = _op_table(lib, rec.calls, 4, 2, (MFMA_PEAK_TFLOPS["bf16"] / 2 if split else MFMA_PEAK_TFLOPS["fp32"]) * 1e12)
        line["kernels"] = tab[:6]
    if split:
        # measured live, on the trained-for-a-few-steps weights of this run: the SAME network evaluated in both arithmetics
        # (the oracle-anchored bound is tests/test_fp32x3_gpu.py; this is the run's own check that the split path is what ran)
        # (the mode is the module's own: an eval COPY in exact fp32 next to the live fp32x3 trainer, no switch flipped)
        import copy
        from rdst_amd import metrics
        net32.eval()
        net_exact = copy.deepcopy(net32).set_compute_dtype("fp32")
        with torch.no_grad():
            y3 = net32(x).float().cpu().numpy()
            y1 = net_exact(x).float().cpu().numpy()
        del net_exact
        net32.train()
        tg = tgt.float().cpu().numpy()
        line["vs_exact_fp32"] = {"out_max_abs_diff": float(abs(y3 - y1).max()),
                                 "psnr_db": [round(metrics.psnr(tg, y3), 6), round(metrics.psnr(tg, y1), 6)],
                                 "abs_dpsnr_db": float(abs(metrics.psnr(tg, y3) - metrics.psnr(tg, y1))),
                                 "what": "forward of this line's network on the step's batch in fp32x3 and in exact fp32 (eval mode, same weights), PSNR against the synthetic targets"}
    return line


def _op_key(n):
    # one op, two entry points: the Linear backward with / without a second addend; window attention with / without the
    # forward's row statistics kept for the backward (window 16)
    return {"rdst_ln_linear_bwd2": "rdst_ln_linear_bwd", "rdst_wattn_fwd_lse": "rdst_wattn_fwd",
            "rdst_wattn_bwd_lse": "rdst_wattn_bwd"}.get(n, n)


def _attn_c(n, a):
    """The channel count C among the positional arguments of a window-attention entry point."""
    return a[{"rdst_swin_attn_fwd": 21, "rdst_wattn_fwd": 10, "rdst_wattn_fwd_lse": 9, "rdst_wattn_bwd": 15, "rdst_wattn_bwd_lse": 16}[n]]


def _op_table(lib, recorded, elt, reps, mfma_peak):
    groups = {}
    for n, a in recorded:
        groups.setdefault(_op_key(n), []).append((n, a))
    table = []
    for n, calls in groups.items():
        ms = _replay_calls(lib, calls, reps)
        by = [_alg(cn, a, elt) for cn, a in calls]
        nbytes, flops = sum(b for b, _ in by) / len(calls), sum(f for _, f in by) / len(calls)
        hbm, mf = nbytes / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), flops / (ms * 1e-3) / mfma_peak
        table.append({"op": n, "launches_per_step": len(calls), "avg_us": round(1e3 * ms, 2),
                      "ms_per_step": round(ms * len(calls), 3), "alg_bytes": int(nbytes), "alg_flops": int(flops),
                      "bound": "hbm" if hbm >= mf else "mfma", "frac_of_peak": round(max(hbm, mf), 4)})
    table.sort(key=lambda r: -r["ms_per_step"])
    return table


def op_tables(lib, recorded, elt, reps, dtype_name, net, x, ws16=False):
    """`kernels`, `roofline` (K1) and `roofline_bwd` (K2) of the line, all from the launches recorded during capture."""
    out = {}
    mfma_peak = MFMA_PEAK_TFLOPS[dtype_name] * 1e12
    groups = {}
    for n, a in recorded:
        groups.setdefault(_op_key(n), []).append((n, a))
    table = _op_table(lib, recorded, elt, reps, mfma_peak)
    out["kernels"] = {"how": "all launches of a C-ABI entry point recorded while the step graph was captured, replayed back to "
                             "back from one HIP graph on the step's own operands (cold: a step's activations are GBs), HIP "
                             "events around the replays; frac_of_peak = max(alg bytes / 8 TB/s, alg FLOPs / dense MFMA peak) / time",
                      "top": table[:8]}
    # ---- roofline of the window-attention forward kernel: K8 (the fused attention half of a Swin block: LayerNorm + qkv ->
    # window attention -> proj + shortcut) where the step runs it, else K1 (window 16, fp32) -------------------------
    fused = "rdst_swin_attn_fwd" in groups
    fa, ci = ("rdst_swin_attn_fwd", 21) if fused else ("rdst_wattn_fwd", 10)
    k1 = groups[fa]
    k1_ms = next(r["avg_us"] for r in table if r["op"] == fa) * 1e-3
    k1_bytes = sum(_alg(n, a, elt)[0] for n, a in k1) / len(k1)
    per_c = {}
    for C in sorted({_attn_c(n, a) for n, a in k1}):
        sub = [(n, a) for n, a in k1 if _attn_c(n, a) == C]
        ms = _replay_calls(lib, sub, reps)
        bts = _alg(sub[0][0], sub[0][1], elt)[0]
        per_c[f"C{C}"] = {"avg_launch_us": round(1e3 * ms, 2), "frac": round(bts / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
    # inside the step: the forward pass as a HIP graph with these launches and with them left out.  The outputs of a left-out launch
    # (buffers of the graph's own pool, never written by anybody else first) are zero-filled ONCE after the capture, so everything
    # downstream — LayerNorm, GELU, the next blocks — runs on finite data in every replay (an uninitialised buffer could hold NaNs).
    in_step = None
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemset2D.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t]

        def outputs(n, a):   # (pointer, pitch, row bytes, rows) of everything the call writes (rows of a wider buffer: only the call's columns)
            if n == "rdst_swin_attn_fwd":
                M, C = a[18] * a[19] * a[20], a[21]
                return [(a[9], a[10] * elt, 3 * C * elt, M), (a[11], a[12] * elt, C * elt, M), (a[13], a[14] * elt, C * elt, M), (a[15], 8, 8, M)]
            if n == "rdst_wattn_fwd_lse":   # (qkv, ld, table, out, ld_out, nlse, B, H, W, C, heads, ...)
                M, C = a[6] * a[7] * a[8], a[9]
                return [(a[3], a[4] * elt, C * elt, M), (a[5], 4 * a[10], 4 * a[10], M)]
            if n == "rdst_wattn_fwd":       # (qkv, ld, table, mask, mask_nw, out, ld_out, B, H, W, C, ...)
                M, C = a[7] * a[8] * a[9], a[10]
                return [(a[5], a[6] * elt, C * elt, M)]
            raise RuntimeError(f"no output map for {n}")

        def fwd_graph(skip):
            with torch.no_grad():
                net(x)
                torch.cuda.synchronize()
                gph = torch.cuda.CUDAGraph()
                with Recorder(lib) as r2:
                    r2.skip = skip
                    with torch.cuda.graph(gph):
                        net(x)
            if skip is not None:
                torch.cuda.synchronize()
                for n, a in r2.calls:
                    if _op_key(n) == skip:
                        for ptr, pitch, wb, rows in outputs(n, a):
                            if ptr and hip.hipMemset2D(ptr, pitch, 0, wb, rows) != 0:
                                hip.hipGetLastError()   # (not sticky: the measurements after this one go on)
                                raise RuntimeError("hipMemset2D failed")
                torch.cuda.synchronize()
            return gph
        g_full, g_skip = fwd_graph(None), fwd_graph(fa)
        diffs = sorted(_timed_replay(g_full, 15) - _timed_replay(g_skip, 15) for _ in range(4))   # interleaved
        in_step = 0.5 * (diffs[1] + diffs[2]) / len(k1)
        del g_full, g_skip
    except Exception as e:  # noqa: BLE001 - measurement aid only
        print(f"bench.py: in-step K1 timing unavailable ({type(e).__name__}: {e})", file=sys.stderr)
    if ws16:
        traffic, tsrc = _pmc_traffic("wattn16_fwd_kernel", ["wattn16_mfma.hip", "wattn_hd.h"])
        # round 5: one head per workgroup (wattn16_bwd1_kernel) for every head dim; the pair kernels remain for the calls without the
        # forward's statistics
        t2, t2src = _pmc_traffic("wattn16_bwd1_kernel", ["wattn16_mfma.hip", "wattn_hd.h"])
        if t2 is None:
            t2, t2src = _pmc_traffic("wattn16_bwd3_kernel", ["wattn16_mfma.hip", "wattn_hd.h"])
        kname, bound_note = "rdst_wattn_fwd (K1, window 16: wattn16_fwd_kernel)", "vector ALU (256 x 256 x 6 exponentials per window); priced against HBM as north_star asks"
    else:
        if fused:
            traffic, tsrc = _pmc_traffic("swinattn_fwd_kernel", ["swinattn_fwd.hip", "wattn_hd.h"])
        else:
            traffic, tsrc = _pmc_traffic("wattn_fwd_hd_kernel", ["wattn_mfma_hd.hip", "wattn_hd.h"])
        # K2 is two kernels since round 4 (wattn_bwd_mfma.hip, RDST_K2_DMA = 6): C = 60 on wattn_bwd_hd_kernel, C = 90 / 120 on
        # wattn_bwd_pair_kernel; the per-launch traffic of the family is the launch-weighted mean of the two collections
        ta, tasrc = _pmc_traffic("wattn_bwd_hd_kernel", ["wattn_bwd_mfma_hd.hip", "wattn_hd.h"])
        tb, tbsrc = _pmc_traffic("wattn_bwd_pair_kernel", ["wattn_bwd_pair.hip", "wattn_hd.h"])
        n60 = sum(1 for n, a in groups["rdst_wattn_bwd"] if _attn_c(n, a) == 60)
        nall = len(groups["rdst_wattn_bwd"])
        if ta is not None and tb is not None and nall:
            t2 = int((n60 * ta + (nall - n60) * tb) / nall)
            t2src = f"{n60} launches x wattn_bwd_hd_kernel ({ta} B) + {nall - n60} x wattn_bwd_pair_kernel ({tb} B): {tasrc}"
        else:
            t2, t2src = None, (tasrc if ta is None else tbsrc)
        kname, bound_note = (("rdst_swin_attn_fwd (K8 = LayerNorm + qkv -> window attention (K1's arithmetic) -> proj + shortcut in one "
                              "launch; algorithmic bytes 6 C elt + 8 per token: x in; qkv, attention output, x1, statistics out)")
                             if fused else "rdst_wattn_fwd (K1, window attention forward)"), None
    ach = k1_bytes / (k1_ms * 1e-3) / 1e9
    out["roofline"] = {"kernel": kname, "id": "K8" if fused else ("K1-ws16" if ws16 else "K1"), "bound": "hbm",
                       "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                       "launches": len(k1), "avg_launch_us": round(1e3 * k1_ms, 2),
                       "how": "COLD: the step's launches of this kernel (every width, shifted and not) replayed back to back on "
                              "the step's own buffers, HIP events on the launch stream; in_step = (forward graph with them) - "
                              "(forward graph without: their outputs zero-filled once after the capture) / launches",
                       "per_shape": per_c, "algorithmic_bytes_per_launch_avg": int(k1_bytes)}
    if bound_note:
        out["roofline"]["note"] = bound_note
    if in_step and in_step > 0:
        a2 = k1_bytes / (in_step * 1e-3) / 1e9
        out["roofline"]["in_step"] = {"avg_launch_us": round(1e3 * in_step, 2), "achieved": round(a2, 1),
                                      "frac": round(a2 / HBM_PEAK_GBS, 4)}
    k2r = next(r for r in table if r["op"] == "rdst_wattn_bwd")
    k2 = groups["rdst_wattn_bwd"]
    per_c2 = {}
    for C in sorted({_attn_c(n, a) for n, a in k2}):
        sub = [(n, a) for n, a in k2 if _attn_c(n, a) == C]
        ms = _replay_calls(lib, sub, reps)
        bts = _alg(sub[0][0], sub[0][1], elt)[0]
        per_c2[f"C{C}"] = {"avg_launch_us": round(1e3 * ms, 2), "frac": round(bts / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
    out["roofline_bwd"] = {"kernel": "rdst_wattn_bwd (K2)", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "achieved": round(k2r["alg_bytes"] / (k2r["avg_us"] * 1e-6) / 1e9, 1),
                           "frac": round(k2r["alg_bytes"] / (k2r["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                           "avg_launch_us": k2r["avg_us"], "per_shape": per_c2, "traffic": t2, "traffic_source": t2src}
    return out


if __name__ == "__main__":
    main()
