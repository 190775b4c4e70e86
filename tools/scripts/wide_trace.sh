# kernel statistics of the quiet chain with more than 64 blocks (N=4096, C=100): tools/wide_probe.py under rocprofv3
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/wide_tr -- python3 $R/tools/wide_probe.py ${1:-4096} ${2:-100} ${3:-400} > $R/gpurun_out/wide_probe.out 2> $R/gpurun_out/wide_tr.err
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/wide_tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open("$R/gpurun_out/wide_trace.txt", "w")
# the last 300 dispatches: the quiet phase with both models running
sel = rows[-300:]
agg = collections.OrderedDict()
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(r["Kernel_Name"][:70], [0, 0.0]); a[0] += 1; a[1] += d
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
out.write("last %d dispatches span %.1f us\n" % (len(sel), span))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.write("  %-70s n=%5d mean %9.2f us total %5.1f%% of span\n" % (k, n, t / n, 100 * t / span))
out.close()
PY
cat $R/gpurun_out/wide_probe.out $R/gpurun_out/wide_trace.txt
