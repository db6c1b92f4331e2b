# usage: bash tools/prof_tool.sh <tag> <python tool> [args]: rocprofv3 kernel stats of one tools/ script -> gpurun_out/<tag>/stats.txt
export TMPDIR=/tmp
export PYTHONPATH=$PWD
R=$PWD
T=$1; shift
mkdir -p gpurun_out/$T
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -o p -- python3 $R/"$@" > $R/gpurun_out/$T/out.txt 2> $R/gpurun_out/$T/err.txt
cd $R
f=$(find gpurun_out/$T/prof -name '*kernel_stats.csv' | head -1)
python3 tools/stats.py $f 12 > gpurun_out/$T/stats.txt
rm -rf gpurun_out/$T/prof
cat gpurun_out/$T/stats.txt
