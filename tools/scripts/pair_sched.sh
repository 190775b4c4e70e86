#!/bin/bash
# Pair scheduling of the two models' resident scans at N=65536, K=64: variants, then the kernel timeline of the default.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "GG_PAIR_PASS=1" "GG_PAIR_PASS=0" "GG_PAIR_SCHED=0" "GG_PAIR_DELAY_US=20" "GG_PAIR_DELAY_US=8"; do
  a=$(env $cfg timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1)
  b=$(env $cfg timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1)
  echo "$cfg: $a | $b"
done
bash tools/scripts/big_trace.sh > gpurun_out/big_trace_run.txt 2>&1
grep -v rocclr gpurun_out/big_trace.txt | head -30
