#!/bin/bash
# Round-5 artifacts in one GPU-box visit: bench lines, rocprofv3 stats of the same commands, timeline, kernel order, gradient
# errors.  usage: tools/gpu_round_r05.sh [bench prof tl trace grads mol dp]
WHAT=${@:-bench prof tl trace grads mol dp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp
cd $R
for w in $WHAT; do
case $w in
bench)
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
  timeout 900 python bench.py --full --no_cpu_baseline --no_configs45 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"; cut -c1-200 $O/bench_full.json;;
prof)
  cd /tmp
  timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof -o run -- python3 $R/bench.py --no_cpu_baseline --no_configs45 > $O/prof_bench.json 2> $O/prof.log; echo "prof rc=$?"
  timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_full -o run -- python3 $R/bench.py --full --no_cpu_baseline --no_configs45 > $O/prof_full_bench.json 2> $O/prof_full.log; echo "prof full rc=$?"
  cd $R
  cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/default_bench_kernel_stats.csv
  cp $(find $O/prof_full -name "*kernel_stats.csv" | head -1) $O/full_bench_kernel_stats.csv;;
tl)
  timeout 300 python tools/probes/step_timeline.py --bucket > $O/step_timeline_device_stamps.txt 2>/dev/null; tail -32 $O/step_timeline_device_stamps.txt
  timeout 300 python tools/probes/step_timeline.py --bucket --full > $O/step_timeline_device_stamps_full.txt 2>/dev/null;;
trace)
  cd /tmp
  timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o run -- python3 $R/bench.py --no_cpu_baseline --no_configs45 --steps 12 --warmup 6 > $O/trace_bench.json 2> $O/trace.log; echo "trace rc=$?"
  cd $R
  python tools/trace_step.py $O/trace > $O/step_kernel_order_under_rocprof.txt 2>&1; head -6 $O/step_kernel_order_under_rocprof.txt
  rm -rf $O/trace;;
grads)
  timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -q -s -k "bs256 or bucket_step_matches" 2>&1 | grep -i "rel\|err\|worst\|passed\|failed" > $O/gradient_error_numbers.txt; cut -c1-300 $O/gradient_error_numbers.txt;;
mol)
  timeout 300 python tools/bench_escore.py > $O/escore_mol_microbench.txt 2>&1; tail -6 $O/escore_mol_microbench.txt
  timeout 300 python tools/bench_sampler.py 1000 > $O/sampler_bench.txt 2>&1; tail -2 $O/sampler_bench.txt
  bash tools/ab_mol.sh > $O/ab_score_kernel_mol_vs_ops.txt 2>&1; cat $O/ab_score_kernel_mol_vs_ops.txt;;
dp)
  MSDE_DP_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 --no_cpu_baseline --no_configs45 > $O/bench_dp2_gloo_one_gpu.json 2> $O/bench_dp2.err; echo "dp2 rc=$?"; cut -c1-300 $O/bench_dp2_gloo_one_gpu.json
  timeout 600 python3 bench.py --debug_dp_path --no_cpu_baseline --no_configs45 > $O/bench_rccl_1rank.json 2> $O/bench_rccl.err; echo "rccl 1 rank rc=$?"; cut -c1-200 $O/bench_rccl_1rank.json;;
esac
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
