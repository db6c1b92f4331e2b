#!/usr/bin/env python3
"""Collect hardware counters for the kernels of ONE training step of bench.py and write profiles/pmc_latest.json.

Run on the GPU box (through gpurun), from the repo root:   python3 tools/pmc_collect.py [tag [config [merge.json]]]
(`merge.json`: an earlier collection whose OTHER kernels are carried over, e.g. e1 + ws16 + e1_hrl into one pmc_latest.json)

Method (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots"): one rocprofv3 run PER counter set, `--kernel-trace --pmc`
only (no other trace domain), the program itself behind `--`:
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -o p -- python3 bench.py --steps 1 --warmup 0
              --graph 0 --no-roofline --no-cpu-baseline
FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots).  On gfx950 FETCH_SIZE tallies 64 B per 128-B request of a
wide streaming read: it is DOUBLED here before it is compared with byte counts; WRITE_SIZE is exact for 16-B stores.
Both are reported by rocprofv3 in KiB.  bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the
launches of the step (for K1 / K2 that is the step's own mix of C = 60 / 90 / 120, shifted and not).
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16 and SIMD), summed over the chip; with
GRBM_GUI_ACTIVE (sum over the 8 XCDs) the matrix-pipe utilisation of a kernel is
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8).
Each entry carries the sha256 of the kernel's source files so bench.py only quotes it for the sources it was measured on.
"""
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]]
# kernel-name substring -> (key in the JSON, source files of that kernel)
KERNELS = {
    "swinattn_fwd_kernel": ("swinattn_fwd_kernel", ["swinattn_fwd.hip", "wattn_hd.h"]),
    "wattn_fwd_hd_kernel": ("wattn_fwd_hd_kernel", ["wattn_mfma_hd.hip", "wattn_hd.h"]),
    "wattn_bwd_hd_kernel": ("wattn_bwd_hd_kernel", ["wattn_bwd_mfma_hd.hip", "wattn_hd.h"]),
    "wattn_bwd_pair_kernel": ("wattn_bwd_pair_kernel", ["wattn_bwd_pair.hip", "wattn_hd.h"]),
    "conv3_kernel": ("conv3_kernel", ["conv3_mfma.hip"]),
    "conv3_wgrad_kernel": ("conv3_wgrad_kernel", ["conv3_wgrad.hip"]),
    "lin_mfma_kernel": ("lin_mfma_kernel", ["linear_mfma.hip"]),
    "mlp_fwd_kernel": ("mlp_fwd_kernel", ["mlp_mfma.hip"]),
    "mlp_bwd_kernel": ("mlp_bwd_kernel", ["mlp_mfma.hip"]),
    "lnlin_bwd_kernel": ("lnlin_bwd_kernel", ["mlp_mfma.hip"]),
    "lnlin3_bwd_kernel": ("lnlin3_bwd_kernel", ["lnlin3_mfma.hip"]),
    "lin3_kernel": ("lin3_kernel", ["lin3_mfma.hip"]),
    "mlp3_fwd_kernel": ("mlp3_fwd_kernel", ["mlp3_mfma.hip"]),
    "wattn16_fwd_kernel": ("wattn16_fwd_kernel", ["wattn16_mfma.hip", "wattn_hd.h"]),
    "wattn16_bwd_kernel": ("wattn16_bwd_kernel", ["wattn16_mfma.hip", "wattn_hd.h"]),
    "wattn16_bwd3_kernel": ("wattn16_bwd3_kernel", ["wattn16_mfma.hip", "wattn_hd.h"]),
    "wattn16_bwd1_kernel": ("wattn16_bwd1_kernel", ["wattn16_mfma.hip", "wattn_hd.h"]),
    "lin3x_kernel": ("lin3x_kernel", ["lin3x_mfma.hip"]),
    "lnlin3x_bwd_kernel": ("lnlin3x_bwd_kernel", ["lnlin3x_mfma.hip"]),
    "conv3x_kernel": ("conv3x_kernel", ["conv3x_mfma.hip"]),
    "conv3x_wgrad_kernel": ("conv3x_wgrad_kernel", ["conv3x_wgrad.hip"]),
    "wattn_fwd_mfma_kernel": ("wattn_fwd_mfma_kernel", ["wattn_mfma.hip"]),
    "wattn_bwd_mfma_kernel": ("wattn_bwd_mfma_kernel", ["wattn_bwd_mfma.hip"]),
    "batched_sum_kernel": ("batched_sum_kernel", ["reduce_batch.hip"]),
    "uconv_halo_kernel": ("uconv_halo_kernel", ["uconv_mfma.hip"]),
    "uconv_kernel": ("uconv_kernel", ["uconv_mfma.hip"]),
}
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "e1"   # bench.py --config (ws16: the window-16 kernels); "x3": e1 in --dtype fp32x3


def source_hash(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "rdst_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def run_pass(counters, outdir):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", outdir, "-o", "p", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--graph", "0", "--no-roofline",
           "--no-cpu-baseline", *(["--config", "e1", "--dtype", "fp32x3"] if CONFIG == "x3" else ["--config", CONFIG])]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:
        print(r.stderr[-2000:], file=sys.stderr)
        raise SystemExit(f"rocprofv3 pass {counters} failed rc={r.returncode}")
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {outdir}")
    return files[0]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "pmc"
    scratch = os.path.join(ROOT, "gpurun_out", tag)
    agg = {}   # key -> per full kernel name -> counter -> [values]
    for i, counters in enumerate(PASSES):
        path = run_pass(counters, os.path.join(scratch, f"pass{i}"))
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                for sub, (key, _) in KERNELS.items():
                    if sub in name and (sub not in ("conv3_kernel", "conv3x_kernel") or "wgrad" not in name):
                        short = re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0]
                        agg.setdefault(key, {}).setdefault(short, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    out = {"command": "rocprofv3 --kernel-trace --pmc <one set per run> -- python3 bench.py --steps 1 --warmup 0 --graph 0 "
                      "--no-roofline --no-cpu-baseline", "passes": PASSES, "kernels": {}}
    for key, variants in agg.items():
        files = next(v[1] for v in KERNELS.values() if v[0] == key)
        ent = {"source_hash": source_hash(files), "variants": {}}
        tot = {}
        for short, ctrs in variants.items():
            v = {c: {"launches": len(x), "avg": sum(x) / len(x)} for c, x in ctrs.items()}
            ent["variants"][short] = v
            for c, x in ctrs.items():
                tot.setdefault(c, []).extend(x)
        mean = {c: sum(x) / len(x) for c, x in tot.items()}
        ent["launches"] = len(next(iter(tot.values())))
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            ent["fetch_KiB_raw"] = mean["FETCH_SIZE"]
            ent["write_KiB"] = mean["WRITE_SIZE"]
            ent["bytes_per_launch"] = int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024)
            ent["how"] = "(2 x FETCH_SIZE + WRITE_SIZE) KiB, averaged over the step's launches; separate --pmc passes"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean.get("GRBM_GUI_ACTIVE"):
            ent["mfma_util"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * mean["GRBM_GUI_ACTIVE"] / 8.0)
            ent["lds_bank_conflict_frac"] = (mean.get("SQ_LDS_BANK_CONFLICT", 0.0) / mean["SQ_LDS_IDX_ACTIVE"]
                                             if mean.get("SQ_LDS_IDX_ACTIVE") else None)
        out["kernels"][key] = ent
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        with open(sys.argv[3]) as fh:
            old = json.load(fh)
        for key, ent in old.get("kernels", {}).items():
            out["kernels"].setdefault(key, ent)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    dst = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for key, ent in out["kernels"].items():
        print(key, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ent.items() if k not in ("variants",)})
    print("wrote", dst, "- copy it to profiles/pmc_latest.json (and a per-round name) to have bench.py quote it")


if __name__ == "__main__":
    main()
