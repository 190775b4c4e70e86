#!/bin/bash
# eigenmodel sweep (configs[2]) in the serial form and with the auxiliary draws overlapped with the feature scan, at several
# numbers of compute units per XCD kept free of the draws
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 0" "1 1" "1 2" "1 4" "1 8"; do
  set -- $cfg
  echo "GG_EIG_OVERLAP=$1 GG_EIG_RESERVE=$2: $(GG_EIG_OVERLAP=$1 GG_EIG_RESERVE=$2 python bench.py --config eigen --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.1f sweeps/s  %.3f ms" % (d["value"], d["ms_per_step"]))')"
done
