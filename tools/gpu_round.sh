#!/bin/bash
# One GPU-box visit: parity tests, bench lines, RCCL 1-rank log, rocprofv3 stats + PMC passes.
# usage: tools/gpu_round.sh <tag> [tests|bench|prof|pmc|dp ...]   (default: all)
TAG=${1:-r02}; shift
WHAT=${@:-tests bench dp prof pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_dp.py > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/gputest.log; tail -5 $O/gputest.log;;
gemm)
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k gemm_ex > $O/gemm_test.log 2>&1; echo "gemm test rc=$?"; tail -15 $O/gemm_test.log
  timeout 600 python tools/bench_gemm_ex.py > $O/gemm_bench.log 2>&1; cat $O/gemm_bench.log;;
head)
  timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "dense_head or sde3d2d or full_pretrain or losscurve_through" > $O/head_test.log 2>&1; echo "head test rc=$?"; tail -40 $O/head_test.log;;
plan)
  timeout 900 python -m pytest tests/test_gpu_plan.py -q -x > $O/plan_test.log 2>&1; echo "plan test rc=$?"; tail -40 $O/plan_test.log;;
trace)
  cd /tmp
  timeout 900 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o run -- python3 $R/bench.py --no_cpu_baseline --steps 12 --warmup 6 $TRACE_ARGS > $O/trace_bench.json 2> $O/trace.log; echo "trace rc=$?"
  cd $R
  python tools/trace_step.py $O/trace > $O/trace_step.txt 2>&1; head -12 $O/trace_step.txt
  rm -rf $O/trace;;
cfpipe)
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "cfconv" > $O/cf_test.log 2>&1; echo "cf test rc=$?"; tail -8 $O/cf_test.log
  for d in 1 0; do MSDE_CFBWD_PIPE=$d timeout 300 python tools/bench_cfconv_bwd.py 2>&1 | grep "dbg=\|fwd "; done | tee $O/cfpipe.log;;
cfdbg)
  for d in 0 1 2 4 7 8 16 32 64 120 127 128 255; do MSDE_CFBWD_DBG=$d timeout 300 python tools/bench_cfconv_bwd.py 2>&1 | grep dbg=; done | tee $O/cfdbg.log;;
md17)
  timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "md17" > $O/md17_test.log 2>&1; echo "md17 test rc=$?"; tail -40 $O/md17_test.log;;
dptest)
  timeout 1200 python -m pytest tests/test_gpu_dp.py -q -x > $O/dptest.log 2>&1; echo "dptest rc=$?"; tail -60 $O/dptest.log;;
bench)
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench.json
  timeout 600 python bench.py --full --no_cpu_baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"; cut -c1-330 $O/bench_full.json;;
dp)
  timeout 600 python bench.py --debug_dp_path --no_cpu_baseline > $O/bench_dp1.json 2> $O/bench_dp1.err; echo "dp rc=$?"; cut -c1-330 $O/bench_dp1.json; tail -3 $O/bench_dp1.err
  timeout 600 python bench.py --debug_dp_path --full --no_cpu_baseline > $O/bench_dp1_full.json 2> $O/bench_dp1_full.err; echo "dp full rc=$?"; cut -c1-330 $O/bench_dp1_full.json;;
prof)
  cd /tmp
  timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof -o run -- python3 $R/bench.py --no_cpu_baseline > $O/prof_bench.json 2> $O/prof.log; echo "prof rc=$?"
  timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_full -o run -- python3 $R/bench.py --full --no_cpu_baseline > $O/prof_full_bench.json 2> $O/prof_full.log; echo "prof full rc=$?"
  cd $R;;
pmc)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/pmc_$c -o run -- python3 $R/tools/prof_kernels.py $PMC_ARG > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  done
  timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/pmc_SQ -o run -- python3 $R/tools/prof_kernels.py $PMC_ARG > $O/pmc_SQ.log 2>&1; echo "pmc SQ rc=$?"
  timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_SQ2 -o run -- python3 $R/tools/prof_kernels.py $PMC_ARG > $O/pmc_SQ2.log 2>&1; echo "pmc SQ2 rc=$?"
  cd $R
  python tools/pmc_summary.py $O/pmc_counters.json '{"N": 3588, "E_r": 49090, "batch": "make_batch(256, seed=0)"}' $(find $O -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log
  find $O -name "*counter_collection.csv" -delete;;
esac
done
# only summaries travel back: drop per-dispatch traces and rocprof's databases
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
