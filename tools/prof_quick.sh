# usage: bash tools/prof_quick.sh <tag> : rocprofv3 kernel stats of 5 training steps only -> gpurun_out/<tag>/per_step.txt
export TMPDIR=/tmp
R=$PWD
T=${1:-profq}
mkdir -p gpurun_out/$T
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench_prof.json 2> $R/gpurun_out/$T/prof.err
cd $R
f=$(find gpurun_out/$T/prof -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/$T/kernel_stats.csv
rm -rf gpurun_out/$T/prof
python3 tools/stats.py gpurun_out/$T/kernel_stats.csv > gpurun_out/$T/per_step.txt
