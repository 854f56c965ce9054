"""Print rocprofv3 kernel stats (run_kernel_stats.csv under a directory) for kernels whose name contains any of the
given substrings: name, calls, average us, percentage.  usage: python tools/kstats.py <dir> <substr> [<substr> ...]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
keys = sys.argv[2:]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if not keys or any(k in n for k in keys):
        print(f'{n.split("(")[0][-46:]:46s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:8.1f} us {float(r["Percentage"]):6.2f} %')
