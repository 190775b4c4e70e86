#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -k "not eigen and not latent" > gpurun_out/mfma_at_size.log 2>&1 || { tail -30 gpurun_out/mfma_at_size.log; exit 1; }
tail -2 gpurun_out/mfma_at_size.log
for cfg in "GG_PAIR_ADJ16=1" "GG_PAIR_ADJ16=0"; do
  a=$(env $cfg timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1)
  b=$(env $cfg timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1)
  echo "$cfg: $a | $b"
done
bash tools/scripts/big_trace.sh > gpurun_out/big_trace_run.txt 2>&1
grep -v rocclr gpurun_out/big_trace.txt | head -16
