"""Summarise rocprofv3 --pmc counter_collection CSVs (any number of passes) into profiles/<tag>_pmc_counters.json:
per kernel (name substring match), per counter: launches, median / min / max of the per-dispatch value (summed over
XCDs / instances, as rocprofv3 reports one row per instance).

    python tools/pmc_summary.py <output json> <shape json> <csv> [<csv> ...]

FETCH_SIZE / WRITE_SIZE are in KB; HBM-side bytes per launch follow MI355X_MICROARCH.md: FETCH_SIZE x 2 for kernels
that read with 16-B-per-lane coalesced loads (gfx950 counts 128-B requests as 64 B), raw for other widths
(uncalibrated), WRITE_SIZE as is."""
import csv, json, statistics, sys, collections, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["cfconv_fused_fwd_kernel", "cfconv_fused_bwd_w_pipe_kernel", "cfconv_aggregate_bwd_x_kernel", "dense_edge_layer",
           "dense_node", "dense_mlp", "dense_pair", "gemm_f32_mfma_kernel", "edge_attention_fwd", "gin_aggregate_fwd"]
WIDE_READS = {"cfconv_aggregate_bwd_x_kernel", "gin_aggregate_fwd", "edge_attention_fwd"}     # float4 row gathers


def main():
    dst, shape = sys.argv[1], json.loads(sys.argv[2])
    vals = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    grid = {}
    for path in sys.argv[3:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                for k in KERNELS:
                    if k in name:
                        key = f'{k}[grid={r["Grid_Size"]}]'
                        vals[key][r["Counter_Name"]][(path, r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {"_note": __doc__.split("\n\n")[-1].replace("\n", " "), "shape": shape}
    for k, counters in sorted(vals.items()):
        ent = {}
        for c, d in sorted(counters.items()):
            v = sorted(d.values())
            ent[c] = {"launches": len(v), "median": statistics.median(v), "min": v[0], "max": v[-1]}
        base = k.split("[")[0]
        if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
            f_kb, w_kb = ent["FETCH_SIZE"]["median"], ent["WRITE_SIZE"]["median"]
            mult = 2 if base in WIDE_READS else 1
            ent["traffic_bytes"] = int((mult * f_kb + w_kb) * 1024)
            ent["traffic_bytes_fetch_doubled"] = int((2 * f_kb + w_kb) * 1024)
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in ent and "SQ_BUSY_CYCLES" in ent:
            ent["mfma_mops_f32_per_busy_cycle"] = ent["SQ_INSTS_VALU_MFMA_MOPS_F32"]["median"] / max(ent["SQ_BUSY_CYCLES"]["median"], 1)
        out[k] = ent
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1)[:6000])


if __name__ == "__main__":
    main()
