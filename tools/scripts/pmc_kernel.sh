# per-kernel mean of a few PMC counters: bash tools/scripts/pmc_kernel.sh NAME "CTR1 CTR2 ..." -- python3 prog args...
# (counters in a run of their own: no trace domains next to --pmc)
set -e
NAME=$1; CTRS=$2; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$NAME
( cd $R && rocprofv3 --pmc $CTRS --output-format csv -d /tmp/pmc_$NAME -- "$@" > $R/gpurun_out/pmc_$NAME.out 2> $R/gpurun_out/pmc_$NAME.err )
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_$NAME/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("$R/gpurun_out/pmc_$NAME.txt", "w")
for k, d in sorted(acc.items(), key=lambda kv: -len(list(kv[1].values())[0])):
    n = len(list(d.values())[0])
    if n < 5: continue
    out.write("%-70s n=%5d " % (k, n) + " ".join("%s=%.4g" % (c, sum(v[len(v)//4:]) / max(1, len(v[len(v)//4:]))) for c, v in sorted(d.items())) + "\n")
out.close()
PY
cat $R/gpurun_out/pmc_$NAME.out | tail -3; cat $R/gpurun_out/pmc_$NAME.txt
