cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02w; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/p1 -o run -- python3 $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $O/p1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/p2 -o run -- python3 $GRAFT_REPO_ROOT/tools/gemm_pmc.py > $O/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02w'
for f in sorted(glob.glob(O+'/p*/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    # dispatch order: first 20 gemm_ex = full, next 20 = no loads
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    order=[]
    for r in rows:
        if 'gemm_ex' not in r['Kernel_Name']: continue
        d=int(r['Dispatch_Id'])
        if d not in order: order.append(d)
        per[d][r['Counter_Name']]+=float(r['Counter_Value'])
    order.sort()
    for name,ids in (('full',order[5:20]),('noload',order[25:40])):
        agg=collections.defaultdict(float)
        for d in ids:
            for c,v in per[d].items(): agg[c]+=v/len(ids)
        print(name, {c: round(v) for c,v in sorted(agg.items())})
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
