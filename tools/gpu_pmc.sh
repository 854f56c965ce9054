#!/bin/bash
# PMC passes (separate runs per counter set; --kernel-trace only beside --pmc) for the dominant kernels
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${ROUND:-r06}pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for m in ${MODES:-gemm step infer}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --output-format csv --kernel-trace --pmc $c -d $O/pmc_${m}_$c -o run -- python3 $R/tools/prof_kernels.py $m > $O/pmc_${m}_$c.log 2>&1; echo "pmc $m $c rc=$?"
  done
  timeout 400 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/pmc_${m}_SQ -o run -- python3 $R/tools/prof_kernels.py $m > $O/pmc_${m}_SQ.log 2>&1; echo "pmc $m SQ rc=$?"
  timeout 400 rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_${m}_SQ2 -o run -- python3 $R/tools/prof_kernels.py $m > $O/pmc_${m}_SQ2.log 2>&1; echo "pmc $m SQ2 rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_counters.json $(find $O -name "*counter_collection.csv") > $O/pmc_summary.txt 2>&1; tail -60 $O/pmc_summary.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
