#!/bin/bash
# The round's artifacts in one GPU-box visit (copied into profiles/ as <ROUND>_*): bench lines (default, --full), rocprofv3 stats of
# the same commands, device-stamp timelines, the kernel order of one step, parity numbers, the data-parallel lines.
# usage: ROUND=r06 tools/gpu_round.sh [tests bench prof tl trace parity dp]
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
WHAT=${@:-tests bench prof tl trace parity dp}
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r06}
O=$R/gpurun_out/$ROUND
mkdir -p $O
export TMPDIR=/tmp
cd $R
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -q -x -m gpu > $O/gputest.log 2>&1; echo "tests rc=$?"; tail -n 4 $O/gputest.log | cut -c1-300;;
bench)
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
  timeout 900 python bench.py --full --no_cpu_baseline --no_configs45 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"; cut -c1-200 $O/bench_full.json;;
prof)
  cd /tmp
  timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof -o run -- python3 $R/bench.py --no_cpu_baseline --no_configs45 --no_config2 > $O/prof_bench.json 2> $O/prof.log; echo "prof rc=$?"
  timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_full -o run -- python3 $R/bench.py --full --no_cpu_baseline --no_configs45 > $O/prof_full_bench.json 2> $O/prof_full.log; echo "prof full rc=$?"
  cd $R
  cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/default_bench_kernel_stats.csv
  cp $(find $O/prof_full -name "*kernel_stats.csv" | head -1) $O/full_bench_kernel_stats.csv;;
tl)
  timeout 300 python tools/probes/step_timeline.py --bucket > $O/step_timeline_device_stamps.txt 2>/dev/null; tail -34 $O/step_timeline_device_stamps.txt
  timeout 300 python tools/probes/step_timeline.py --bucket --full > $O/step_timeline_device_stamps_full.txt 2>/dev/null; tail -12 $O/step_timeline_device_stamps_full.txt;;
trace)
  cd /tmp
  timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o run -- python3 $R/bench.py --no_cpu_baseline --no_configs45 --no_config2 --steps 12 --warmup 6 > $O/trace_bench.json 2> $O/trace.log; echo "trace rc=$?"
  cd $R
  python tools/trace_step.py $O/trace > $O/step_kernel_order_under_rocprof.txt 2>&1; head -6 $O/step_kernel_order_under_rocprof.txt
  rm -rf $O/trace;;
parity)
  rm -f gpurun_out/parity_numbers.txt
  timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -q -s -m gpu -k "bs256 or loss_curve or bucket_step_matches" > $O/parity_tests.log 2>&1; echo "parity rc=$?"
  cp gpurun_out/parity_numbers.txt $O/parity_numbers.txt; cut -c1-300 $O/parity_numbers.txt;;
dp)
  bash tools/dp2_gloo_smoke.sh $ROUND; echo "dp2 rc=$?"
  timeout 600 python3 bench.py --debug_dp_path --no_cpu_baseline --no_configs45 > $O/bench_rccl_1rank.json 2> $O/bench_rccl.err; echo "rccl 1 rank rc=$?"; cut -c1-200 $O/bench_rccl_1rank.json;;
esac
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
