#!/bin/bash
# PMC counters of ONE kernel (name substring) over a short bench run; prints per-counter medians.  usage: tools/pmc_one.sh <substr> [bench args]
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
K=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc1
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d /tmp/pmc1/$(echo $set | cut -c1-12) -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --steps 3 --warmup 2 "$@" > /dev/null 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, sys, collections, statistics
k = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            vals[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
for c, d in sorted(vals.items()):
    v = sorted(d.values())
    print(f"{c:28s} launches {len(v):4d}  median {statistics.median(v):14.0f}")
PY
