"""print (name, calls, average us) of the kernels whose name contains argv[2] from a rocprofv3 kernel_stats csv (argv[1])."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
