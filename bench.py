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

Extra objects on the line (N = 1, rank 0):
  roofline     — the window-attention forward kernel (K1): algorithmic bytes (4*C*elt per token per launch,
                 DESIGN.md) / its average duration INSIDE the step, vs 8 TB/s HBM.  The duration is measured
                 with HIP events on the launch stream by difference: the forward pass as a HIP graph with its
                 48 K1 launches, minus the same graph with a memset in place of each (the memset is timed and
                 added back).  `cold_replay` is the same 48 launches (real operands, distinct buffers) replayed
                 back to back from a graph inside one event pair — cold HBM reads, the conservative figure.
                 `traffic` = HBM bytes per launch from the PMC counters (profiles/README.md).
  roofline_bwd — K2 the same way (cold replay only).
  cpu_baseline — the CPU oracle (oracle/rdst_oracle.py, a port) timed on the host cores on a bounded
                 sample (batch 4) of the same workload.
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per K1 launch from the PMC counters (profiles/README.md: separate --pmc passes, FETCH_SIZE
# doubled for wide loads as the guide prescribes), averaged over the step's 48 launches; None = not collected
K1_TRAFFIC_BYTES_PER_LAUNCH = 96272384   # profiles/r01p_pmc_wattn_fetch_write.json: (2*FETCH + WRITE) KiB averaged over C = 60/90/120


def build_net(device, dtype):
    from rdst_amd.networks.rdst_variations import RDSTSR
    torch.manual_seed(0)                       # seeded default init, as SURVEY.md §8d prescribes
    net = RDSTSR(**E1)
    net.to(device).train().set_compute_dtype(dtype)
    return net


def _cpu_baseline_worker(threads, sample_batch, timed):
    """Runs in a child process: the oracle (a CPU port of the reference algorithm), fwd + L1 + bwd."""
    torch.set_num_threads(threads)
    from oracle import rdst_oracle as O
    cfg = O.CFG_E1
    sd = O.make_weights(cfg, 0)
    lay = O.state_dict_layout(cfg)
    sd = {k: (v.clone().requires_grad_(True) if lay[k][2] not in ("index", "mask", "shift_w", "shift_b") else v)
          for k, v in sd.items()}
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(sample_batch, 1, 64, 64, generator=g)
    tgt = torch.rand(sample_batch, 1, 256, 256, generator=g)
    best = None
    for it in range(1 + timed):
        t0 = time.perf_counter()
        y = O.rdstsr_forward(x, sd, cfg)
        F.l1_loss(y, tgt).backward()
        dt = time.perf_counter() - t0
        if it > 0:
            best = dt if best is None else min(best, dt)
    print(json.dumps({"best_s": best}), flush=True)


def cpu_baseline(threads=16, sample_batch=4, timed=3, timeout_s=150):
    """Bounded CPU baseline next to the GPU number.  16 threads: on the 256-CPU GPU-box host the
    oracle is FASTEST there (measured 1.7 s/step at 16 threads, 2.9 s at 32, 5.7 s at 64: the ops are
    small and OpenMP fork/join dominates beyond that), so this is the best CPU figure, not a handicap."""
    import subprocess
    threads = max(1, min(threads, os.cpu_count() or 1))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), str(sample_batch), str(timed)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        best = json.loads(r.stdout.strip().splitlines()[-1])["best_s"]
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "patches/s", "cores": threads, "kind": "port",
                "sample": f"failed or exceeded {timeout_s}s: {type(e).__name__}"}
    return {"value": round(sample_batch / best, 3), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"oracle/rdst_oracle.py fwd+L1+bwd, RDST-E1 x4, batch {sample_batch} of 1x64x64 fp32, "
                      f"best of {timed} after 1 warm-up, torch CPU, {threads} threads"}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-baseline-worker":
        _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=3, help="replay passes over the captured K1/K2 launches")
    ap.add_argument("--no-roofline", action="store_true", help="skip the K1/K2 measurements (clean per-step kernel profiles)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs a torch.distributed.run launch with that many ranks",
                  file=sys.stderr)
            sys.exit(2)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    from rdst_amd import dp, ops, optim
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    net = build_net(device, dtype)
    dp.broadcast_parameters(net)
    bucket = dp.FlatGradBucket(net.parameters())
    # utils/optim.py:30-53 with the ini's hyper-parameters; one fused HIP launch over the flat buffers
    opt = optim.FlatAdam(bucket.params, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, bucket=bucket)
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    x = torch.rand(B, 1, 64, 64, generator=g).to(device)
    tgt = torch.rand(B, 1, 256, 256, generator=g).to(device)
    loss_buf = torch.zeros((), device=device)

    def fwd_bwd():
        bucket.detach_grads()       # autograd assigns fresh grads (no per-parameter accumulate kernels) ...
        y = net(x)
        loss = F.l1_loss(y, tgt)
        loss_buf.copy_(loss.detach())
        loss.backward()
        bucket.gather()             # ... which are flattened into the one all-reduce / Adam bucket

    def step_eager():
        fwd_bwd()
        bucket.all_reduce_mean()
        opt.step()

    graph = None
    if args.graph:
        # HIP graph of forward+backward (static shapes, no host sync inside); the collective and the
        # optimizer stay outside so RCCL is free to use its own streams
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step_eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fwd_bwd()
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"bench.py: graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
        if graph is not None and not bucket.check_views():
            graph = None

    def step():
        if graph is not None:
            graph.replay()
            bucket.all_reduce_mean()
            opt.step()
        else:
            step_eager()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss_val = float(loss_buf.item())

    out = {
        "metric": "SR patches/sec fwd+bwd, RDST-E1 x4 64->256",
        "value": round(world * B * args.steps / elapsed, 3),
        "unit": "patches/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "RDST-E1 x4 (RDST_E1_OASIS_example_SRx4.ini), 1x64x64 LR patches -> 256x256, "
                               "step = fwd + L1 + bwd + flat-bucket grad all-reduce + Adam",
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                   "hip_graph": graph is not None, "grad_bucket_bytes": bucket.nbytes},
        "loss": round(loss_val, 6),
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        # ---- roofline of the window-attention forward kernel (K1), HIP events on the launch stream ----
        # One eager step with a capturing KernelTimer: every K1 / K2 launch of the step (its real operands,
        # 48 launches each, C = 60/90/120, shifted and not) is kept and then replayed back to back inside
        # ONE event pair on the launch stream.  (An event pair around a single 20-50 us launch reads
        # 5-10 us high against rocprofv3's kernel durations; the per-launch brackets are reported too.)
        kt = ops.KernelTimer(capture=True)
        ops.set_kernel_timer(kt)
        step_eager()
        torch.cuda.synchronize()
        ops.set_kernel_timer(None)
        f, b = kt.summary("fwd"), kt.summary("bwd")
        reps = max(1, args.roofline_steps)
        f_ms, b_ms = kt.replay("fwd", reps), kt.replay("bwd", reps)
        fbytes, bbytes = f["bytes"] / f["launches"], b["bytes"] / b["launches"]
        # In the training step K1 reads a qkv that the preceding Linear has just written (partly still in the
        # Infinity Cache); the replay reads cold buffers.  The in-step duration is the per-launch bracket of the
        # instrumented step minus what a bracket adds, measured on the same launches (bracket_overhead).
        b_step_ms = b["total_ms"] / b["launches"]

        # K1 inside the step, by difference: the forward pass as a HIP graph with and without its 48 K1 launches
        # (the zero-fill that stands in for K1 is timed on its own and added back), one event pair around 10 replays.
        def fwd_graph():
            with torch.no_grad():
                net(x)                                   # warm-up outside capture
                torch.cuda.synchronize()
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph):
                    net(x)
            return gph

        def timed(gph, n=10):
            gph.replay()
            torch.cuda.synchronize()
            t0e, t1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0e.record()
            for _ in range(n):
                gph.replay()
            t1e.record()
            torch.cuda.synchronize()
            return t0e.elapsed_time(t1e) / n

        f_step_ms = None
        try:
            g_full = fwd_graph()
            ops.SKIP_K1_FOR_TIMING = True
            g_skip = fwd_graph()
            ops.SKIP_K1_FOR_TIMING = False
            zt = [torch.empty(B, 64, 64, c, device=device, dtype=dtype) for c in (60, 90, 120)]
            torch.cuda.synchronize()
            z0, z1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            z0.record()
            for _ in range(16):
                for z in zt:
                    z.zero_()
            z1.record()
            torch.cuda.synchronize()
            zero_ms = z0.elapsed_time(z1) / 48            # average stand-in memset
            diffs = []
            for _ in range(4):                          # interleaved, so clock drift cancels
                diffs.append(timed(g_full, 15) - timed(g_skip, 15))
            diffs.sort()
            f_step_ms = 0.5 * (diffs[1] + diffs[2]) / f["launches"] + zero_ms   # median of four
            del g_full, g_skip
        except Exception as e:  # noqa: BLE001 - measurement aid only
            ops.SKIP_K1_FOR_TIMING = False
            print(f"bench.py: in-step K1 timing unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        if not f_step_ms or f_step_ms <= 0:
            f_step_ms = f_ms
        ach = fbytes / (f_step_ms * 1e-3) / 1e9
        ach_cold = fbytes / (f_ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "rdst_wattn_fwd (K1, window attention forward)", "bound": "hbm",
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": K1_TRAFFIC_BYTES_PER_LAUNCH,
                           "launches": f["launches"], "avg_launch_us": round(1e3 * f_step_ms, 2),
                           "how": "in the step: (forward graph with K1) - (forward graph with a memset in its place) over the "
                                  "48 launches, HIP events around 10 graph replays each; cold_replay = the same 48 "
                                  "launches replayed back to back on cold buffers",
                           "avg_launch_us_single_bracket": round(1e3 * f["total_ms"] / f["launches"], 2),
                           "cold_replay": {"achieved": round(ach_cold, 1), "frac": round(ach_cold / HBM_PEAK_GBS, 4),
                                           "avg_launch_us": round(1e3 * f_ms, 2), "launches": f["launches"] * reps},
                           "algorithmic_bytes_per_launch_avg": int(fbytes)}
        achb = bbytes / (b_ms * 1e-3) / 1e9
        out["roofline_bwd"] = {"kernel": "rdst_wattn_bwd (K2)", "bound": "hbm", "achieved": round(achb, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achb / HBM_PEAK_GBS, 4),
                               "avg_launch_us": round(1e3 * b_ms, 2),
                               "avg_launch_us_single_bracket": round(1e3 * b_step_ms, 2)}
        del kt
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
