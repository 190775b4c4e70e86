#!/bin/bash
# configs[3] shape (N=65536, K=64) with every scan-kernel variant under _build/variants/ and the default library:
# both models together (the sweep that counts), the weight model alone with its per-round timers.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in graphistician_amd/_build/libgraphgibbs.so graphistician_amd/_build/variants/lib_*.so; do
  n=$(basename $f .so)
  b=$(GG_LIB_PATH=$R/$f timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1)
  w=$(GG_LIB_PATH=$R/$f GG_ROUND_DBG=1 timeout -k 10 120 python tools/one_model.py w 65536 64 20 2>&1 | tail -2)
  echo "== $n"; echo "$b"; echo "$w"
done
