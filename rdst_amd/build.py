"""Build librdst_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m rdst_amd.build          # incremental; --force rebuilds everything
    python -m rdst_amd.build --debug  # librdst_hip_dbg.so with -DRDST_DEBUG (ablation switches, in-kernel stamps,
                                      # environment overrides: tools/ only; load it with RDST_HIP_LIB=<path>)

One object per .hip file (so edits rebuild in seconds), linked into rdst_amd/librdst_hip.so, which
stays in-tree: it is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "librdst_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]



def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force: bool = False, verbose: bool = True, debug: bool = False) -> str:
    OBJ = os.path.join(HERE, "csrc", "_obj_dbg" if debug else "_obj")
    LIB = os.path.join(HERE, "librdst_hip_dbg.so" if debug else "librdst_hip.so")
    FLAGS = globals()["FLAGS"] + (["-DRDST_DEBUG"] if debug else [])
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "rdst_hip.h"))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + hdrs, obj):
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[rdst_amd.build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(objs, LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, debug="--debug" in sys.argv))
