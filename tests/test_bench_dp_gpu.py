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


def test_bench_refuses_mismatched_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
