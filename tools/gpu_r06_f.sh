#!/bin/bash
# Round 6, visit 6: the data-parallel tail (bucket-wise weight gradients + all-reduce), GPU suite, gloo smoke line, 1-rank RCCL line.
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f
mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu > $O/gputest.log 2>&1; echo "tests rc=$?"; tail -n 6 $O/gputest.log | cut -c1-300
bash tools/dp2_gloo_smoke.sh r06f; echo "dp2 rc=$?"
timeout 600 python3 bench.py --debug_dp_path --no_cpu_baseline --no_configs45 > $O/bench_rccl_1rank.json 2> $O/bench_rccl.err; echo "rccl 1 rank rc=$?"; cut -c1-300 $O/bench_rccl_1rank.json
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06f/bench_rccl_1rank.json") if l.startswith("{")][-1])
print("RCCL 1 rank: headline", j["ms_per_step"], "parts", {k: j["dp_step_parts_us"].get(k) for k in ("graph_us", "allreduce_us", "adam_us", "allreduce_alone_us")})
print("             full    ", j["config2_full"]["ms_per_step"], {k: j["config2_full"]["dp_step_parts_us"].get(k) for k in ("graph_us", "allreduce_us", "adam_us")})
PY
timeout 900 python tools/ab_set.py 2 overlap=--dp nooverlap=moleculesde_amd.pretrain:DP_OVERLAP=False,--dp 2>&1 | tail -4
