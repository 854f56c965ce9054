cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "Name:[ ]*[A-Za-z0-9_]*" | sed 's/Name:[ ]*//' | sort -u | tr '\n' ' ' > $O/counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/pmcg_$tag -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm_rs.py 3588x300x300 3588x300x600 > $O/pmcg_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03"
for f in sorted(glob.glob(O+"/pmcg_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]+" grid="+r.get("Grid_Size","")
        if "rsa" in k or "Cijk" in k or "gemm_ex" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
