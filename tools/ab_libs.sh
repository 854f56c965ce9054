# A/B of library builds (tools/_build/libmsde_*.so) on the headline bench, alternating on ONE box.  usage: tools/ab_libs.sh ROUNDS name1 name2 ...
cd $GRAFT_REPO_ROOT
R=$1; shift
cp moleculesde_amd/csrc/libmsde_hip.so /tmp/libmsde_orig.so
for i in $(seq 1 $R); do
  for v in "$@"; do
    cp tools/_build/libmsde_$v.so moleculesde_amd/csrc/libmsde_hip.so
    ms=$(python bench.py --no_cpu_baseline --no_configs45 --no_pipeline --no_bf16x3 --steps 300 $BENCH_ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$v] $ms"
  done
done
cp /tmp/libmsde_orig.so moleculesde_amd/csrc/libmsde_hip.so
