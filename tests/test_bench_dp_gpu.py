"""bench.py's N > 1 path, executed for real: two ranks of `bench.py --gpus 2 --backend gloo` (the command line the driver
uses, with gloo instead of RCCL so that both ranks may share the one GPU of this box, RDST_BENCH_ONE_GPU=1) — process-group
init, parameter broadcast, graph-captured step, flat-bucket all-reduce, MAX-over-ranks timing, one JSON line from rank 0.
Everything but the RCCL transport of the driver's 8-GPU run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
def test_bench_two_ranks_gloo_prints_one_line():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RDST_BENCH_ONE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                                       "--warmup", "1", "--no-roofline", "--no-cpu-baseline"], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=800) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    lines1 = [l for l in outs[1][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and len(lines1) == 0          # rank 0 prints THE line, rank 1 nothing
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "patches/s"
    assert d["config"]["hip_graph"] is True and d["config"]["backend"] == "gloo"
    assert d["param_sync"] is True                          # both replicas hold identical parameters after the steps
    assert d["value"] > 0 and d["loss"] == d["loss"] and 0.0 < d["loss"] < 10.0
    assert abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) <= 0.01 * d["value"]


def _torchrun_bench(extra, timeout=800):
    """The DOCUMENTED command (bench.py:4-6), through the launcher: a fresh child of the pytest process, two ranks on the one
    GPU of this box over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RDST_BENCH_ONE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
           "--warmup", "1", "--no-roofline", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_through_torch_distributed_run():
    d = _torchrun_bench([])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["param_sync"] is True and d["config"]["hip_graph"] is True
    # a scaling record explains itself: what the process group really was and what the collective cost
    assert d["world_size_seen"] == 2 and d["backend"] == "gloo" and d["bucket_bytes"] == 4 * 4464961
    assert d["all_reduce_ms"] > 0.0
    assert abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) <= 0.01 * d["value"]


@pytest.mark.timeout(900)
def test_bench_config5_two_ranks_through_torch_distributed_run():
    """BASELINE configs[4] (RDST-HRL: the seg-UNet label-hr loss in the backward path) on two ranks: equal parameters after the
    steps, one all-reduced loss, and the replicated loss network ran the same number of BatchNorm batches on every rank."""
    d = _torchrun_bench(["--config", "e1_hrl", "--batch", "2"])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["param_sync"] is True
    assert d["config"]["unet_dtype"] == "fp32x3" and d["config"]["hip_graph"] is True
    lo, hi = d["unet_bn_batches_tracked"]
    assert lo == hi == 2 * (2 + 1 + 3)      # SR + HR pass per step: 2 eager steps, 1 warm-up and 3 timed replays (a capture executes nothing)
    assert d["loss"] == d["loss"] and 0.0 < d["loss"] < 10.0


def test_bench_refuses_mismatched_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


@pytest.mark.timeout(900)
def test_bench_one_rank_over_rccl():
    """The RCCL transport of row (e), executed: `bench.py --gpus 1 --backend nccl --force-pg` through the launcher (a fresh
    child: the process group is initialised before anything else touches the GPU) — a 1-rank NCCL(=RCCL) process group bound
    to the device, the parameter broadcast, HIP-graph capture and replays with the NCCL watchdog alive, the AVG all-reduce of
    the 17.9 MB flat bucket, FlatAdam.  Everything the 8-GPU run does except moving bytes over xGMI."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                          "RDST_BENCH_ONE_GPU")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-pg",
           "--steps", "3", "--warmup", "1", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp1"
    assert d["world_size_seen"] == 1 and d["backend"] == "nccl" and d["config"]["backend"] == "nccl"
    assert d["bucket_bytes"] == 4 * 4464961 and d["all_reduce_ms"] > 0.0
    assert d["config"]["hip_graph"] is True and d["param_sync"] is True
    assert d["loss"] == d["loss"] and 0.0 < d["loss"] < 10.0
