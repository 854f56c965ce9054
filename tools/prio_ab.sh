# A/B of compile-time wave priorities, both builds on ONE box.  usage: tools/prio_ab.sh "<flagsA>" "<flagsB>" [rounds]
cd $GRAFT_REPO_ROOT
SRC=$(ls moleculesde_amd/csrc/*.hip | tr '\n' ' ')
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-unused-value $1 $SRC -o /tmp/lib_$2.so 2>/dev/null; }
build "$1" A & build "$2" B & wait
cp moleculesde_amd/csrc/libmsde_hip.so /tmp/lib_orig.so
for i in $(seq 1 ${3:-3}); do
  for v in A B; do
    cp /tmp/lib_$v.so moleculesde_amd/csrc/libmsde_hip.so
    ms=$(python bench.py --no_cpu_baseline --steps 60 $BENCH_ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "$v $ms"
  done
done
cp /tmp/lib_orig.so moleculesde_amd/csrc/libmsde_hip.so
