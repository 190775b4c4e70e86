# kernel-level picture of a chain that keeps changing labels (bench chain seed 2, N=4096, K=16)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GG_BENCH_CHAIN_SEED=2 GG_ROUND_DBG=1 python3 $R/bench.py --no-cpu --no-extras --steps 100 > $R/gpurun_out/mix_dbg.json 2> $R/gpurun_out/mix_dbg.err
grep "scan dbg" $R/gpurun_out/mix_dbg.err | tail -8
GG_BENCH_CHAIN_SEED=2 python3 $R/bench.py --no-cpu --no-extras --steps 200 > $R/gpurun_out/mix_plain.json 2> /dev/null
python3 -c "
import json; d=json.load(open('$R/gpurun_out/mix_plain.json')); print('seed 2:', d['value'], d['ms_per_step'], d['config'].get('label_changes_in_timed_region'))"
if [ -n "$MIX_PROF" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mix -- python3 $R/bench.py --no-cpu --no-extras --steps 200 > $R/gpurun_out/mix_prof.json 2> $R/gpurun_out/mix_prof.err
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_mix/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-60s calls %6s avg %9.1f ns  total %6.2f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
fi
