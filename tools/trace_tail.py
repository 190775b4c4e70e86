"""Steady-state per-kernel durations from a rocprofv3 kernel trace (last n dispatches)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tail = rows[-n:]
d = collections.defaultdict(list)
for r in tail:
    d[r['Kernel_Name'].split('(')[0][:40]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in d.items():
    print("%-42s n=%5d avg %8.1f us  med %8.1f  max %8.1f" % (k, len(v), sum(v) / len(v) / 1e3, sorted(v)[len(v) // 2] / 1e3, max(v) / 1e3))
span = (int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])) / 1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tail) / 1e3
print("span %.1f us busy %.1f us" % (span, busy))
k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for r in tail[n // 2:n // 2 + k]:
    print(r['Kernel_Name'].split('(')[0][:30], (int(r['Start_Timestamp']) - int(tail[n // 2]['Start_Timestamp'])) / 1e3,
          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X'), r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Scratch_Size'))
