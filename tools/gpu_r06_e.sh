#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e
mkdir -p $O
rm -f gpurun_out/parity_numbers.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -q -s -x -m gpu -k "bs256 or loss_curve or bucket_step_matches" > $O/parity_tests.log 2>&1; echo "parity tests rc=$?"; tail -3 $O/parity_tests.log | cut -c1-200
cp gpurun_out/parity_numbers.txt $O/parity_numbers.txt; cat $O/parity_numbers.txt | cut -c1-400
S=moleculesde_amd.hip
timeout 1500 python tools/ab_set.py 3 --full grp6= grp3=$S:CFCONV_BWD_GROUP=3 grp2=$S:CFCONV_BWD_GROUP=2 r5mode=$S:CFCONV_FILTER_MULTI=False,$S:CFCONV_BWD_GROUP=1 2>&1 | tee $O/ab_multi_step_full.txt
timeout 1500 python tools/ab_set.py 3 grp3=$S:CFCONV_BWD_GROUP=3 r5mode=$S:CFCONV_FILTER_MULTI=False,$S:CFCONV_BWD_GROUP=1 grp6= 2>&1 | tee $O/ab_multi_step.txt
