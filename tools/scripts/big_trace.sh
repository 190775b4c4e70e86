# kernel timeline of the N=65536, K=64 chain (last sweeps of a short run)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/big_tr -- python3 $R/bench.py --nodes 65536 --blocks 64 --steps 6 --warmup 2 --burnin 12 --no-cpu --no-extras > /dev/null 2> $R/gpurun_out/big_tr.err
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/big_tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if not r["Kernel_Name"].startswith("k_wait")]
out = open("$R/gpurun_out/big_trace.txt", "w")
first_ll = next((i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_loglik")), len(rows))
for r in rows[max(0, first_ll - 90):first_ll]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%-46s q%-3s start %12.1f us  dur %9.1f us\n" % (r["Kernel_Name"][:46], r.get("Queue_Id", "?"), s / 1e3, (e - s) / 1e3))
out.close()
PY
tail -70 $R/gpurun_out/big_trace.txt
