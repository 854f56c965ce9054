#!/bin/bash
# Round 6, visit 4: CFConv work hoisted out of the layer chain (all blocks' filter rows / filter-weight gradients in one launch).
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d
mkdir -p $O
timeout 1200 python -m pytest tests -q -x -m gpu > $O/gputest.log 2>&1; echo "tests rc=$?"; tail -n 6 $O/gputest.log | cut -c1-300
S=moleculesde_amd.hip
timeout 2400 python tools/ab_set.py 3 new= r5lib=lib:tools/_build/libmsde_pair_r5.so,$S:CFCONV_FILTER_MULTI=False,$S:CFCONV_BWD_GROUP=1 \
   fwdonly=$S:CFCONV_BWD_GROUP=1 bwdonly=$S:CFCONV_FILTER_MULTI=False grp3=$S:CFCONV_BWD_GROUP=3 grp2=$S:CFCONV_BWD_GROUP=2 \
   bpw3=$S:CFCONV_MULTI_BLOCKS_PER_WG=3 bpw6=$S:CFCONV_MULTI_BLOCKS_PER_WG=6 bpw9=$S:CFCONV_MULTI_BLOCKS_PER_WG=9 2>&1 | tee $O/ab_multi_step.txt
timeout 900 python tools/ab_set.py 2 --full new= r5lib=lib:tools/_build/libmsde_pair_r5.so,$S:CFCONV_FILTER_MULTI=False,$S:CFCONV_BWD_GROUP=1 2>&1 | tee $O/ab_multi_step_full.txt
