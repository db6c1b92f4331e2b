# usage: bash tools/prof_round.sh <tag>   (on the GPU box)  -> gpurun_out/<tag>/*: everything profiles/ needs for a round:
#   bench lines (e1 default, ws16, e1_unetf, e1_hrl), rocprofv3 kernel stats of the training steps (e1, e1_hrl), the kernel
#   trace of the default command reduced to cold-replay vs in-step durations of K1 / K2, and the PMC collection.
export TMPDIR=/tmp
R=$PWD
T=${1:-round}
O=$R/gpurun_out/$T
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err       # (round 4: the default line carries configs[3] / configs[4] too)
python3 bench.py --config ws16 > $O/ws16_bench.json 2>> $O/bench.err
bash tools/prof_cfg.sh $T/e1 > /dev/null 2>&1
bash tools/prof_cfg.sh $T/hrl --config e1_hrl > /dev/null 2>&1
bash tools/prof_cfg.sh $T/ws16 --config ws16 > /dev/null 2>&1
bash tools/prof_cfg.sh $T/x3 --dtype fp32x3 > /dev/null 2>&1    # (round 6: the parity mode's own kernel family)
python3 -m pytest tests/test_fullsize_gpu.py tests/test_segunet_gpu.py tests/test_config5_gpu.py tests/test_modules_gpu.py tests/test_grad_accum_gpu.py tests/test_fp32x3_gpu.py -m gpu -q -s > $O/parity.log 2>&1
cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line > $O/bench_traced.json 2> $O/trace.err
cd $R
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 tools/cold_from_trace.py $f swinattn_fwd_kernel > $O/k8_cold_vs_step.txt
python3 tools/cold_from_trace.py $f wattn_bwd_hd_kernel > $O/k2_cold_vs_step.txt
python3 tools/cold_from_trace.py $f wattn_bwd_pair_kernel >> $O/k2_cold_vs_step.txt
rm -rf $O/trace
python3 tools/pmc_collect.py $T/pmc_e1 e1 > $O/pmc_e1.log 2>&1
python3 tools/pmc_collect.py $T/pmc_ws16 ws16 gpurun_out/$T/pmc_e1_pmc.json > $O/pmc_ws16.log 2>&1
python3 tools/pmc_collect.py $T/pmc_all e1_hrl gpurun_out/$T/pmc_ws16_pmc.json > $O/pmc_hrl.log 2>&1
rm -rf $O/pmc_e1 $O/pmc_ws16 $O/pmc_all
bash tools/prof_tool.sh $T/w16f32 tools/wattn16_bench.py 8 fp32 > /dev/null 2>&1   # the exact-fp32 window-16 kernels at the bench shape
ls -la $O
