cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc=$?"; tail -5 gpurun_out/r03/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "value", d["value"])
r=d["roofline"]; print("roofline", r["frac"], r["avg_launch_us"], r["timing"], r["standalone"])
for k in ("roofline_cfconv_pair_filter","roofline_cfconv_pair_bwd_w","roofline_hbm_message_passing"):
    print(k, d[k]["frac"], d[k]["avg_launch_us"])
f=d["roofline_forward_schnet_sde2d3d"]; print("fwd", f["ms"], f["hbm"]["frac"], f["fp32_flop_floor"]["frac"])
print("c4", d.get("config4_sampler")); print("c5", d.get("config5_md17"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","ms_per_step")})
PY
