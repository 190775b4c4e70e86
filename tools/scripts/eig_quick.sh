#!/bin/bash
# quick check of the eigenmodel after a kernel change: parity at size + the sweep rate (+ chain counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python -m pytest tests/test_eigenmodel.py tests/test_gpu_at_size.py tests/test_golden.py -x -q -m gpu -k "eigen or probit or logistic or golden" > gpurun_out/eigq.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/eigq.log
timeout -k 10 200 python bench.py --config eigen --no-cpu > gpurun_out/eigq_bench.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/eigq_bench.log | head -1
GG_EIG_DBG=1 timeout -k 10 200 python bench.py --config eigen --steps 3 --warmup 1 --no-cpu > gpurun_out/eigq_dbg.log 2>&1; grep "eig chain" gpurun_out/eigq_dbg.log | tail -2
