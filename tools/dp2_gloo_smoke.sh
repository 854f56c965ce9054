#!/bin/bash
# One-GPU smoke run of the `--gpus 2` line: two gloo ranks share the device (RCCL refuses two ranks per device), so the
# numbers are functional only; what is checked is the line's shape (n_gpus, config2_full, the DP step pieces, per-rank spread).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
R=${1:-r06}
mkdir -p gpurun_out/$R
MSDE_DP_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 6 --warmup 2 --no_cpu_baseline --no_configs45 \
    > gpurun_out/$R/bench_dp2_gloo_one_gpu.json 2> gpurun_out/$R/bench_dp2_gloo_one_gpu.err || { echo "dp2 rc=$?"; tail -20 gpurun_out/$R/bench_dp2_gloo_one_gpu.err; exit 1; }
python3 - "$R" <<'PY'
import json, sys
j = json.loads([l for l in open("gpurun_out/%s/bench_dp2_gloo_one_gpu.json" % sys.argv[1]) if l.startswith("{")][-1])
assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2, j
c2, parts = j["config2_full"], j["dp_step_parts_us"]
assert c2["ms_per_step"] > 0 and c2["n_gpus"] == 2
for k in ("graph_us", "allreduce_us", "adam_us", "allreduce_alone_us"):
    assert parts[k] is not None and c2["dp_step_parts_us"][k] is not None, k
assert len(j["ms_per_step_per_rank"]["per_rank"]) == 2
print("dp2 gloo smoke OK: headline %.1f ms, full %.1f ms, allreduce_us %.0f (gloo through the host)" % (j["ms_per_step"], c2["ms_per_step"], parts["allreduce_us"]))
PY
