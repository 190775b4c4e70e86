#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity_sbm.py tests/test_gpu_edge_cases.py -x -q -m gpu > gpurun_out/mfma_parity.log 2>&1 || { tail -30 gpurun_out/mfma_parity.log; exit 1; }
tail -1 gpurun_out/mfma_parity.log
GG_ROUND_DBG=1 timeout -k 10 120 python tools/one_model.py w 65536 64 20 2>&1 | tail -2
GG_ROUND_DBG=1 timeout -k 10 120 python tools/one_model.py adj 65536 64 20 2>&1 | tail -2
timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1
timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1
