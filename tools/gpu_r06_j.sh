#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
ROUND=r06 bash tools/gpu_round.sh bench
ROUND=r06 MODES="step" bash tools/gpu_pmc.sh
