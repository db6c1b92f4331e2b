#!/bin/bash
# usage: bash tools/pmc_step.sh <tag> [bench.py args, e.g. --dtype fp32x3]  -> gpurun_out/<tag>_sq.txt: SQ issue counters of every kernel of one eager e1 training step
export PYTHONPATH=$PWD
T=${1:-sq}
shift
B="python3 bench.py --steps 1 --warmup 0 --graph 0 --no-roofline --no-cpu-baseline $*"
python3 tools/pmc_tool.py "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" _kernel -- $B > gpurun_out/${T}_sq1.txt 2>&1
python3 tools/pmc_tool.py "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" _kernel -- $B > gpurun_out/${T}_sq2.txt 2>&1
python3 - "$T" <<'PY'
import sys, re, collections
T = sys.argv[1]
d = collections.defaultdict(dict)
for f in (f"gpurun_out/{T}_sq1.txt", f"gpurun_out/{T}_sq2.txt"):
    k = None
    for line in open(f):
        if not line.startswith("   "):
            k = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean\s+([\d.]+)", line)
            if m and k:
                d[k][m.group(1)] = float(m.group(3)); d[k]["n"] = int(m.group(2))
print(f"{'kernel':58s} {'n':>4s} {'VALU/wave':>9s} {'cyc/VALU':>8s} {'VALUbusy':>8s} {'MFMAbusy':>8s} {'LDSact':>7s} {'active':>7s} {'wait':>6s} {'stall':>6s}")
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1].get("n", 0)):
    if "SQ_WAVE_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    wc, waves = c["SQ_WAVE_CYCLES"], max(c.get("SQ_WAVES", 1), 1)
    gui = c["GRBM_GUI_ACTIVE"] / 8            # cycles the kernel ran
    print(f"{k[-58:]:58s} {c['n']:4d} {c['SQ_INSTS_VALU'] / waves:9.0f} {4 * c['SQ_ACTIVE_INST_VALU'] / max(c['SQ_INSTS_VALU'], 1):8.2f} "
          f"{4 * c['SQ_ACTIVE_INST_VALU'] / 1024 / gui:8.2f} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui:8.2f} {c['SQ_LDS_IDX_ACTIVE'] / 256 / gui:7.2f} "
          f"{c['SQ_ACTIVE_INST_ANY'] / wc:7.2f} {c['SQ_WAIT_ANY'] / wc:6.2f} {c['SQ_WAIT_INST_ANY'] / wc:6.2f}")
PY
