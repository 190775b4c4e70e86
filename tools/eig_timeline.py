"""Timeline of the last eigenmodel sweep in a rocprofv3 --kernel-trace CSV: when each chunk of auxiliary draws ran and when the
feature scan's launches ran (overlapped form).   python tools/eig_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_glob = max(i for i, r in enumerate(rows) if "k_eig_draw_globals" in r["Kernel_Name"])
prev_glob = max(i for i, r in enumerate(rows[:last_glob]) if "k_eig_draw_globals" in r["Kernel_Name"])
sw = rows[prev_glob + 1:last_glob + 1]
t0 = int(sw[0]["Start_Timestamp"])
print("sweep: %d dispatches, %.1f us" % (len(sw), (int(sw[-1]["End_Timestamp"]) - t0) / 1e3))
nchain = 0
for r in sw:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if "k_eig_chain" in name:
        nchain += 1
        if nchain % 8 != 1:
            continue
        name += " #%d" % nchain
    if "vblocks" in name:
        continue
    print("  %-34s %9.1f -> %9.1f  (%7.1f us)  queue %s" % (name, a, b, b - a, r.get("Queue_Id", "?")))
