# usage: bash tools/prof_trace.sh <tag> <kernel substring> [bench.py args...] -> gpurun_out/<tag>/trace_<substr>.txt : per-launch
# durations (us, launch order, last replayed step only) of the kernels whose name contains the substring
export TMPDIR=/tmp
R=$PWD
T=${1:-trace}
SUB=$2
shift; shift
mkdir -p gpurun_out/$T
cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$T/prof -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/$T/bench_prof.json 2> $R/gpurun_out/$T/prof.err
cd $R
f=$(find gpurun_out/$T/prof -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$SUB" > gpurun_out/$T/trace_$SUB.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = everything after the second-to-last adam launch
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo = adam[-2] + 1 if len(adam) >= 2 else 0
hi = adam[-1]
k = 0
for r in rows[lo:hi]:
    n = r["Kernel_Name"]
    if sys.argv[2] in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{k:4d} {d:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}x{r.get('Grid_Size_Y', '')} {short}")
        k += 1
PY
rm -rf gpurun_out/$T/prof
