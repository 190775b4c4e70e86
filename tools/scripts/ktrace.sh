# kernel statistics of the last dispatches of any command: bash tools/scripts/ktrace.sh NAME LAST -- python3 prog args...
# (program directly after --: no wrapper between rocprofv3 and the process that initialises the GPU)
set -e
NAME=$1; LAST=$2; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$NAME
( cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$NAME -- "$@" > $R/gpurun_out/kt_$NAME.out 2> $R/gpurun_out/kt_$NAME.err )
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/kt_$NAME/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-$LAST:]
agg = collections.OrderedDict()
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(r["Kernel_Name"][:90], [0, 0.0]); a[0] += 1; a[1] += d
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
out = open("$R/gpurun_out/kt_$NAME.txt", "w")
out.write("last %d dispatches span %.1f us\n" % (len(sel), span))
busy = 0.0
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.write("  %-90s n=%5d mean %9.2f us total %5.1f%% of span\n" % (k, n, t / n, 100 * t / span)); busy += t
out.write("  sum of kernel time %.1f%% of span\n" % (100 * busy / span))
out.close()
PY
cat $R/gpurun_out/kt_$NAME.txt
