#!/bin/bash
# kernel-trace statistics of configs[3] on one GPU (N=65536, K=64); run on the GPU box from the repo root
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_big
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_big
echo "start $(date +%T)" > $O/progress.log
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_big -- python3 $R/bench.py --nodes 65536 --blocks 64 --steps 20 --warmup 2 --no-cpu --no-extras > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo "rocprof rc=$? $(date +%T)" >> $O/progress.log
f=$(find /tmp/p_big -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
head -12 $f | cut -c1-200
