"""Summarise rocprofv3 --pmc counter_collection CSVs (one counter per pass) into
profiles/r01_pmc_<COUNTER>_cfconv.csv and profiles/r01_pmc_traffic.json.
Usage: python tools/pmc_summary.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv>"""
import csv, json, statistics, sys, collections, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"cfconv_aggregate_fwd_kernel": "void cfconv_aggregate_fwd_kernel<4>", "cfconv_fused_fwd_kernel": "void cfconv_fused_fwd_kernel<26>"}


def per_kernel(path, counter):
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            for short, full in KERNELS.items():
                if short in name:
                    vals[short][r["Dispatch_Id"]] += float(r["Counter_Value"])     # summed over XCDs / instances
    return {k: sorted(v.values()) for k, v in vals.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    for counter, data in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        with open(os.path.join(ROOT, "profiles", f"r01_pmc_{counter}_cfconv.csv"), "w") as f:
            f.write("kernel,counter,launches,median_value_KB,min,max\n")
            for k, v in data.items():
                f.write(f'"{KERNELS[k]}",{counter},{len(v)},{statistics.median(v)},{min(v)},{max(v)}\n')
    out = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
    fa, wa = statistics.median(fetch["cfconv_aggregate_fwd_kernel"]), statistics.median(write["cfconv_aggregate_fwd_kernel"])
    ff, wf = statistics.median(fetch["cfconv_fused_fwd_kernel"]), statistics.median(write["cfconv_fused_fwd_kernel"])
    out["cfconv_aggregate_fwd_kernel"] = {"fetch_kb_raw": fa, "write_kb": wa, "traffic_bytes": int((2 * fa + wa) * 1024)}
    out["cfconv_fused_fwd_kernel"].update({"fetch_kb_raw": ff, "write_kb": wf, "traffic_bytes": int((ff + wf) * 1024),
                                           "traffic_bytes_if_doubled": int((2 * ff + wf) * 1024)})
    json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
