# rocprofv3 kernel stats of the headline bench under two environment settings.  usage: tools/prof_ab.sh "ENV=a" "ENV=b"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ab; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  export $cfg
  timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/p$i -o run -- python3 $R/bench.py --no_cpu_baseline --no_configs45 --no_bf16x3 --no_pipeline --steps 100 > $O/bench$i.json 2> $O/log$i.txt
  cp $(find $O/p$i -name "*kernel_stats.csv" | head -1) $O/stats$i.csv
  rm -rf $O/p$i
done
