cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline $PROF_ARGS > $O/prof_bench.json 2> $O/prof.log; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:8.1f}us pct={float(r['TotalDurationNs'])/tot*100:5.2f}")
PY
