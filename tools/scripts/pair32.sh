#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -k "cfg3" > gpurun_out/at_size_pair.log 2>&1 || { tail -30 gpurun_out/at_size_pair.log; exit 1; }
tail -1 gpurun_out/at_size_pair.log
for f in graphistician_amd/_build/libgraphgibbs.so graphistician_amd/_build/variants/lib_adjpw4.so; do
  for i in 1 2 3; do GG_LIB_PATH=$R/$f timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1; done
  GG_LIB_PATH=$R/$f GG_ROUND_DBG=1 timeout -k 10 120 python tools/one_model.py adj 65536 64 20 2>&1 | tail -2
done
