"""Print the rows of a rocprofv3 kernel_stats CSV whose kernel name contains any of the given substrings.
usage: python tools/kstats_grep.py <dir or csv> substr [substr ...]"""
import csv, glob, os, sys
src = sys.argv[1]
f = src if os.path.isfile(src) else sorted(glob.glob(src + "/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f"{int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us  ({float(r['Percentage']):5.2f} %)  {r['Name'][:110]}")
