#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -k "cfg3" > gpurun_out/at_size_pair.log 2>&1 || { tail -30 gpurun_out/at_size_pair.log; exit 1; }
tail -1 gpurun_out/at_size_pair.log
for steps in 20 100; do
  timeout -k 10 200 python tools/one_model.py both 65536 64 $steps 2>&1 | tail -1
  GG_PAIR_SCHED=0 timeout -k 10 200 python tools/one_model.py both 65536 64 $steps 2>&1 | tail -1
done
