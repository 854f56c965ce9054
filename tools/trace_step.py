"""Summarise ONE steady-state training step from a rocprofv3 --kernel-trace CSV: kernels in start order with their
queue, start offset and duration; per-queue busy time; idle gaps of the busiest queue.  Run on the GPU box after
`rocprofv3 --kernel-trace --output-format csv`; prints a few KB of text.
usage: python tools/trace_step.py <dir containing *_kernel_trace.csv> [marker kernel substring = adam_chunks]"""
import csv, glob, sys, collections
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_chunks"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# a step = between two consecutive marker kernels (steady state, graph replay): the LAST one that is a headline step --
# built from a raw blob (plan kernels inside) and without the diagnostic time stamps bench.py's roofline leg switches on
def _pick():
    best = None
    for a, b in zip(marks[-2::-1], marks[:0:-1]):
        names = [r["Kernel_Name"] for r in rows[a + 1:b + 1]]
        if len(names) < 50 or any("debug_stamp" in n for n in names):
            continue
        if any("plan_scan" in n for n in names):
            return a, b
        best = best or (a, b)
    return best or (marks[-3], marks[-2])
i0, i1 = _pick()
step = rows[i0 + 1:i1 + 1]
t0 = int(step[0]["Start_Timestamp"])
T = int(step[-1]["End_Timestamp"]) - t0
print(f"step: {len(step)} kernels, {T / 1e3:.1f} us wall")
qkey = "Queue_Id" if "Queue_Id" in step[0] else "Stream_Id"
busy = collections.defaultdict(int)
for r in step:
    busy[r[qkey]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"queue {q}: busy {b / 1e3:.1f} us ({100.0 * b / T:.0f} %), {sum(1 for r in step if r[qkey] == q)} kernels")
# union busy time over all queues (GPU not idle)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
cur_s, cur_e, union = ev[0][0], ev[0][1], 0
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((cur_e - t0, s - cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"GPU busy (any queue) {union / 1e3:.1f} us, idle {(T - union) / 1e3:.1f} us in {len(gaps)} gaps "
      f"(mean {((T - union) / max(len(gaps), 1)) / 1e3:.2f} us)")
mainq = max(busy, key=busy.get)
print("---- kernels in start order (offset us, dur us, queue, grid, name)")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-70:]
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{r[qkey]} {grid:>8}/{wg:<4} {name}")
