#!/bin/bash
# Round-4 artifacts in one GPU-box visit: bench lines, rocprofv3 stats of the same commands, PMC passes, timeline, kernel
# order, GEMM microbench, gradient-error numbers.  usage: tools/gpu_round_r03.sh [bench prof pmc tl trace gemm grads]
WHAT=${@:-bench prof pmc tl trace gemm grads}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
cd $R
for w in $WHAT; do
case $w in
bench)
  timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
  timeout 600 python bench.py --full --no_cpu_baseline --no_configs45 --no_bf16x3 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"; cut -c1-200 $O/bench_full.json;;
prof)
  cd /tmp
  timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof -o run -- python3 $R/bench.py --no_cpu_baseline --no_configs45 --no_bf16x3 > $O/prof_bench.json 2> $O/prof.log; echo "prof rc=$?"
  timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_full -o run -- python3 $R/bench.py --full --no_cpu_baseline --no_configs45 --no_bf16x3 > $O/prof_full_bench.json 2> $O/prof_full.log; echo "prof full rc=$?"
  cd $R
  cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/default_bench_kernel_stats.csv
  cp $(find $O/prof_full -name "*kernel_stats.csv" | head -1) $O/full_bench_kernel_stats.csv;;
pmc)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/pmc_$c -o run -- python3 $R/tools/prof_kernels_r04.py > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  done
  timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/pmc_SQ -o run -- python3 $R/tools/prof_kernels_r04.py > $O/pmc_SQ.log 2>&1; echo "pmc SQ rc=$?"
  timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_SQ2 -o run -- python3 $R/tools/prof_kernels_r04.py > $O/pmc_SQ2.log 2>&1; echo "pmc SQ2 rc=$?"
  cd $R
  python tools/pmc_summary_r04.py $O/pmc_counters.json $(find $O -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1; tail -40 $O/pmc_summary.log
  find $O -name "*counter_collection.csv" -delete;;
tl)
  timeout 300 python tools/probes/step_timeline.py --bucket > $O/step_timeline_device_stamps.txt 2>/dev/null; cat $O/step_timeline_device_stamps.txt;;
trace)
  cd /tmp
  timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o run -- python3 $R/bench.py --no_cpu_baseline --no_configs45 --no_bf16x3 --steps 12 --warmup 6 > $O/trace_bench.json 2> $O/trace.log; echo "trace rc=$?"
  cd $R
  python tools/trace_step.py $O/trace > $O/step_kernel_order_under_rocprof.txt 2>&1; head -6 $O/step_kernel_order_under_rocprof.txt
  rm -rf $O/trace;;
gemm)
  timeout 300 python tools/bench_gemm_t2.py > $O/gemm_t2_microbench.txt 2>&1; cat $O/gemm_t2_microbench.txt;;
grads)
  timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -q -s -k "bs256 or bucket_step_matches" 2>&1 | grep -i "rel\|err\|worst\|passed\|failed" > $O/gradient_error_numbers.txt; cat $O/gradient_error_numbers.txt | cut -c1-300;;
esac
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
