cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 --full 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "full default $(run X=1)"
echo "full GEO=0   $(run MSDE_GEOMETRY_ON_SIDE=0)"
python bench.py --no_cpu_baseline --steps 40 > gpurun_out/r03/bench_default.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print("default ms", d["ms_per_step"], "fwd", d["roofline_forward_schnet_sde2d3d"]["ms"], d["roofline_forward_schnet_sde2d3d"]["hbm"]["frac"], d["roofline_forward_schnet_sde2d3d"]["fp32_flop_floor"]["frac"])
print("stream", d["config"]["stream"])
print("node mlp", d["roofline_dense_head_node_mlp"]["frac"], d["roofline_dense_head_node_mlp"]["us_per_chain"])
PY
