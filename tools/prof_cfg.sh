# usage: bash tools/prof_cfg.sh <tag> [bench.py args...]  -> gpurun_out/<tag>/{kernel_stats.csv,per_step.txt,bench_prof.json}
# rocprofv3 kernel trace + stats of 5 timed training steps of any bench configuration (no roofline replays: in-step launches only)
export TMPDIR=/tmp
R=$PWD
T=${1:-prof}
shift
mkdir -p gpurun_out/$T
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/$T/bench_prof.json 2> $R/gpurun_out/$T/prof.err
cd $R
f=$(find gpurun_out/$T/prof -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/$T/kernel_stats.csv
rm -rf gpurun_out/$T/prof
python3 - "$T" <<'PY' | tee gpurun_out/$T/per_step.txt
import csv,sys
T=sys.argv[1]
rows=list(csv.DictReader(open(f'gpurun_out/{T}/kernel_stats.csv')))
steps=[int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']][0]
print('steps',steps)
for r in rows[:60]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('__hip_bfloat16','bf16').replace('void ','')
    ms=int(r['TotalDurationNs'])/1e6/steps
    print(f"{int(r['Calls'])/steps:6.1f}/step {ms:7.3f} ms/step {float(r['AverageNs'])/1e3:8.1f} us  {n[:120]}")
print('sum of all', sum(int(r['TotalDurationNs']) for r in rows)/1e6/steps)
PY
