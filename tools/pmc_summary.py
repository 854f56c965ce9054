"""rocprofv3 --pmc counter_collection CSVs -> profiles/rNN_pmc_counters.json: per kernel and counter the median per-dispatch
value (summed over instances).  FETCH_SIZE / WRITE_SIZE are in KB; HBM-side bytes per launch follow MI355X_MICROARCH.md:
FETCH_SIZE x 2 (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B; these kernels read with 16-B or 8-B accesses
per lane) + WRITE_SIZE.  usage: python tools/pmc_summary.py <out json> <csv> ...  (one script for every round: rounds 3-5 kept
a copy each)"""
import csv, json, statistics, sys, collections
KEYS = {"gemm_t2_kernel<5, 0": "gemm_t2_kernel<5, 0>[3588x300x300]", "gemm_t2_kernel<5, 2": "gemm_t2_kernel<5, 2>[3588x600x300 bnbwd]",
        "gemm_grouped_wgrad_kernel": "gemm_grouped_wgrad_kernel[--full step, 157 problems]",
        "dense_edge_layer_fwd": "dense_edge_layer_fwd", "dense_edge_layer_bwd": "dense_edge_layer_bwd",
        "escore_mol_fwd_kernel<true": "escore_mol_fwd_kernel<true>", "escore_mol_bwd_kernel": "escore_mol_bwd_kernel",
        "escore_mol_fwd_kernel<false, true": "escore_mol_fwd_kernel<false, true>[sampler: 10 x 14 atoms]",
        "escore_edge_pre_kernel": "escore_edge_pre_kernel[sampler: 1820 edges]", "gemm_small_kernel": "gemm_small_kernel[MD17 step]",
        "cfconv_pair_filter_kernel": "cfconv_pair_filter_kernel", "cfconv_fused_bwd_w_pipe_kernel": "cfconv_fused_bwd_w_pipe_kernel",
        "cfconv_pair_filter_multi_kernel": "cfconv_pair_filter_multi_kernel[6 blocks, one launch]",
        "cfconv_pair_bwd_w_multi_kernel": "cfconv_pair_bwd_w_multi_kernel[filter-weight gradients, blocks per launch as in the --full step]",
        "cfconv_pair_aggregate_kernel": "cfconv_pair_aggregate_kernel"}
vals = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        for k, name in KEYS.items():
            if k in r["Kernel_Name"]:
                vals[name][r["Counter_Name"]][(path, r["Dispatch_Id"])] += float(r["Counter_Value"])
out = {"_note": __doc__.replace("\n", " "), "shape": {"N": 3588, "batch": "make_batch(256, seed=0)"}}
for name, counters in vals.items():
    ent = {}
    for c, d in counters.items():
        v = sorted(d.values())
        ent[c] = {"launches": len(v), "median": statistics.median(v), "min": v[0], "max": v[-1]}
    if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
        ent["traffic_bytes"] = int((2 * ent["FETCH_SIZE"]["median"] + ent["WRITE_SIZE"]["median"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and "GRBM_GUI_ACTIVE" in ent:
        ent["matrix_pipe_busy_frac"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"]["median"] / (1024 * ent["GRBM_GUI_ACTIVE"]["median"] / 8)
    if "SQ_INSTS_VALU" in ent and "SQ_INSTS_MFMA" in ent and ent["SQ_INSTS_MFMA"]["median"] > 0:
        ent["valu_per_mfma"] = ent["SQ_INSTS_VALU"]["median"] / ent["SQ_INSTS_MFMA"]["median"]
    if "SQ_LDS_BANK_CONFLICT" in ent and "SQ_LDS_IDX_ACTIVE" in ent and ent["SQ_LDS_IDX_ACTIVE"]["median"] > 0:
        ent["lds_bank_conflict_frac"] = ent["SQ_LDS_BANK_CONFLICT"]["median"] / ent["SQ_LDS_IDX_ACTIVE"]["median"]
    if "SQ_WAIT_INST_ANY" in ent and "SQ_WAVE_CYCLES" in ent and ent["SQ_WAVE_CYCLES"]["median"] > 0:
        ent["wait_inst_frac_of_wave_cycles"] = ent["SQ_WAIT_INST_ANY"]["median"] / ent["SQ_WAVE_CYCLES"]["median"]
    out[name] = ent
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {kk: (vv if not isinstance(vv, dict) else vv["median"]) for kk, vv in v.items()} for k, v in out.items() if isinstance(v, dict) and k != "shape"}, indent=1))
