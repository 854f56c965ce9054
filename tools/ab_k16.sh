# bf16x3 experiment: K = 32 instruction vs two K = 16 instructions vs the fp32 step, two-stream step, alternating on one box
cd $GRAFT_REPO_ROOT
cp moleculesde_amd/csrc/libmsde_hip.so /tmp/libmsde_orig.so
run() { python bench.py --no_cpu_baseline --no_configs45 --no_pipeline --no_bf16x3 --steps 300 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2; do
  echo "[fp32 two streams] $(run)"
  cp tools/_build/libmsde_k32.so moleculesde_amd/csrc/libmsde_hip.so
  echo "[bf16x3 K=32, two streams] $(MSDE_BF16X3=1 MSDE_BF16X3_TWO_STREAMS=1 run)"
  cp tools/_build/libmsde_k16.so moleculesde_amd/csrc/libmsde_hip.so
  echo "[bf16x3 K=16, two streams] $(MSDE_BF16X3=1 MSDE_BF16X3_TWO_STREAMS=1 run)"
  cp /tmp/libmsde_orig.so moleculesde_amd/csrc/libmsde_hip.so
done
