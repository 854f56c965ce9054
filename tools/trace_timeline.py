"""Analyse a rocprofv3 --kernel-trace CSV of bench.py (hipGraph replay): isolate the last replayed step and
report wall time, per-queue busy time, idle gaps, concurrency and the top kernels of that one step."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# step boundaries: the Adam kernel ends each step
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
print("adam launches", len(idx))
nback = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = idx[-1 - nback] + 1, idx[-nback] + 1
step = rows[lo:hi]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
print(f"kernels in step {len(step)}  wall {(t1 - t0) / 1e3:.1f} us")
# union busy time and concurrency
ev = []
for r in step:
    ev.append((r["s"], 1)); ev.append((r["e"], -1))
ev.sort()
busy = 0; conc_time = collections.Counter(); cur = 0; last = t0
for t, d in ev:
    conc_time[cur] += t - last
    last = t; cur += d
print("time by #kernels in flight (us):", {k: round(v / 1e3, 1) for k, v in sorted(conc_time.items())})
byq = collections.defaultdict(list)
for r in step:
    byq[(r["Queue_Id"], r["Stream_Id"])].append(r)
for q, rs in byq.items():
    b = sum(r["e"] - r["s"] for r in rs)
    print(f"queue/stream {q}: {len(rs)} kernels, busy {b / 1e3:.1f} us, span {(max(r['e'] for r in rs) - min(r['s'] for r in rs)) / 1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r["Kernel_Name"][:60]
    agg[n][0] += 1; agg[n][1] += r["e"] - r["s"]
tot = sum(v[1] for v in agg.values())
print(f"sum of kernel durations {tot / 1e3:.1f} us")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n:60s} x{c:4d} {d / 1e3:8.1f} us  avg {d / c / 1e3:6.1f}")
short = sum(1 for r in step if r["e"] - r["s"] < 5000)
print("kernels < 5 us:", short, " sum", sum(r["e"] - r["s"] for r in step if r["e"] - r["s"] < 5000) / 1e3)
