"""Steady-state timeline summary of a rocprofv3 --kernel-trace CSV: per-kernel mean duration over the last `n` dispatches,
the span they cover and how busy each hardware queue was.

    python tools/trace_gaps.py <..._kernel_trace.csv> [n]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) == 3 else 2000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 4:
    # window between the a-th and the b-th dispatch of the kernel whose name contains argv[2]: python ... csv k_bulk_w 80 230
    key, a, b = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    marks = [int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]]
    lo, hi = marks[a], marks[b]
    rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
    print("window: %d launches of %s" % (b - a, key))
else:
    rows = rows[-n:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
span = t1 - t0
per = defaultdict(list)
queue = defaultdict(int)
for r in rows:
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per[r["Kernel_Name"].split("(")[0][:60]].append(dur)
    queue[r.get("Queue_Id", "?")] += dur
print("last %d dispatches span %.1f us" % (len(rows), span / 1e3))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s n=%5d mean %8.2f us  total %5.1f%% of span" % (k, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / span))
for q, busy in sorted(queue.items()):
    print("  queue %s busy %.1f%%" % (q, 100.0 * busy / span))
