#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity_sbm.py tests/test_gpu_edge_cases.py -x -q -m gpu > gpurun_out/mfma_small_parity.log 2>&1 || { tail -30 gpurun_out/mfma_small_parity.log; exit 1; }
tail -1 gpurun_out/mfma_small_parity.log
GG_SCAN=1 timeout -k 10 300 python tests/fuzz_parity.py 40 7 > gpurun_out/fuzz_small.log 2>&1 || { tail -20 gpurun_out/fuzz_small.log; exit 1; }
tail -1 gpurun_out/fuzz_small.log
for s in 2 4; do
  for v in 1 0; do
    echo "== seed $s GG_MFMA_SMALL=$v"
    GG_MFMA_SMALL=$v GG_ROUND_DBG=1 timeout -k 10 100 python tools/one_model.py w 4096 16 400 $s 2>&1 | tail -2
    GG_MFMA_SMALL=$v timeout -k 10 100 python tools/one_model.py both 4096 16 400 $s 2>&1 | tail -1
  done
done
