#!/bin/bash
# gg_lp_mfma, second pass: parity on the 33..64-block paths, then N=65536 K=64 and N=4096 K=16 with and without the adjacency
# model's queue priority.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity_sbm.py tests/test_gpu_edge_cases.py -x -q -m gpu > $O/mfma_parity.log 2>&1 || { tail -30 $O/mfma_parity.log; exit 1; }
tail -2 $O/mfma_parity.log
timeout -k 10 400 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -k "not eigen and not latent" > $O/mfma_at_size.log 2>&1 || { tail -30 $O/mfma_at_size.log; exit 1; }
tail -2 $O/mfma_at_size.log
for prio in 1 0; do
  echo "== GG_ADJ_PRIO=$prio"
  GG_ADJ_PRIO=$prio timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1
  GG_ADJ_PRIO=$prio timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1
  GG_ADJ_PRIO=$prio timeout -k 10 120 python tools/one_model.py both 4096 16 2000 2>&1 | tail -1
  GG_ADJ_PRIO=$prio timeout -k 10 120 python tools/one_model.py both 4096 16 2000 2>&1 | tail -1
done
GG_ROUND_DBG=1 timeout -k 10 120 python tools/one_model.py w 65536 64 20 2>&1 | tail -2
GG_ROUND_DBG=1 timeout -k 10 120 python tools/one_model.py adj 65536 64 20 2>&1 | tail -2
