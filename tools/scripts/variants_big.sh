#!/bin/bash
# N=65536, K=64 (both models; the weight model alone) and N=4096, K=16 (both) with the default library and every variant
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in graphistician_amd/_build/libgraphgibbs.so graphistician_amd/_build/variants/lib_*.so; do
  n=$(basename $f .so)
  a=$(GG_LIB_PATH=$R/$f timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1)
  b=$(GG_LIB_PATH=$R/$f timeout -k 10 120 python tools/one_model.py both 65536 64 20 2>&1 | tail -1)
  w=$(GG_LIB_PATH=$R/$f timeout -k 10 120 python tools/one_model.py w 65536 64 20 2>&1 | tail -1)
  s=$(GG_LIB_PATH=$R/$f timeout -k 10 120 python tools/one_model.py both 4096 16 2000 2>&1 | tail -1)
  echo "== $n: $a | $b | $w | N=4096: $s"
done
