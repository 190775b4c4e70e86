#!/bin/bash
# kernel-trace statistics of the eigenmodel sweep (configs[2]); run on the GPU box from the repo root
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_eigen
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_eig
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_eig -- python3 $R/bench.py --config eigen --steps 20 --warmup 3 --no-cpu > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo "rocprof rc=$?"
f=$(find /tmp/p_eig -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
head -14 $f | cut -c1-180
