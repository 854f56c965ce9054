cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --output-format csv --kernel-trace -d $O/trace -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --steps 12 --warmup 6 $TRACE_ARGS > $O/trace_bench.json 2> $O/trace.log; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $O/trace > $O/trace_step.txt 2>&1; head -5 $O/trace_step.txt
rm -rf $O/trace
