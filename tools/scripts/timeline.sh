# timeline of the last dispatches of a command: bash tools/scripts/timeline.sh NAME LAST -- python3 prog args...
set -e
NAME=$1; LAST=$2; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$NAME
( cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$NAME -- "$@" > $R/gpurun_out/tl_$NAME.out 2> $R/gpurun_out/tl_$NAME.err )
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/tl_$NAME/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-$LAST:]
t0 = int(sel[0]["Start_Timestamp"])
out = open("$R/gpurun_out/tl_$NAME.txt", "w")
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%-60s q%-3s start %10.1f us  dur %8.1f us  grid %s\n" % (r["Kernel_Name"][:60], r.get("Queue_Id", "?"), s / 1e3, (e - s) / 1e3, r.get("Grid_Size", "?")))
out.close()
PY
cat $R/gpurun_out/tl_$NAME.out; cat $R/gpurun_out/tl_$NAME.txt
