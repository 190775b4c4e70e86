# timeline of the driver's short run (--steps 20 --warmup 5): where the timed region's 0.86 ms go
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/short_tr -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $R/gpurun_out/short.json 2> /dev/null
python3 - <<PY
import csv, glob, json
f = glob.glob("/tmp/short_tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 19 k_sweep_q<3 launches and what surrounds them
idx = [i for i, r in enumerate(rows) if "k_sweep_q<3" in r["Kernel_Name"]]
# clusters of launches less than 200 us apart: the timed region is the one with 19 members
clusters, cur = [], [idx[0]]
for a, b in zip(idx, idx[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) < 200000: cur.append(b)
    else: clusters.append(cur); cur = [b]
clusters.append(cur)
last = [c for c in clusters if len(c) == 19][-1]
lo = last[0] - 5
hi = last[-1] + 7
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-44s q%-3s start %9.1f us  dur %7.1f us" % (r["Kernel_Name"][:44], r.get("Queue_Id", "?"), s / 1e3, (e - s) / 1e3))
d = json.load(open("$R/gpurun_out/short.json"))
print("bench: ms_per_step", d["ms_per_step"], "timed region", d["ms_per_step"] * 20 * 1000, "us")
PY
