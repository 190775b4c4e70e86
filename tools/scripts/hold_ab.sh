#!/bin/bash
# The adaptive scan form's hold rule on the driver's seed mix (N=4096, K=16, 20-sweep calls) and on 200-sweep calls
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "GG_SCAN_HOLD_OLD=1" "GG_SCAN_HOLD_OLD=0" "GG_SCAN_HOLD_MIN=8" "GG_SCAN_HOLD_MIN=32" "GG_SCAN_HOLD_Q=16" "GG_SCAN_HOLD_LONG=32"; do
  for steps in 20 200; do
    env $cfg timeout -k 10 200 python bench.py --gpus 1 --steps $steps --warmup 5 --no-cpu --no-extras > gpurun_out/hold_tmp.json 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/hold_tmp.json") if l.startswith("{")][-1])
print("$cfg steps=$steps: %.0f sweeps/s | " % d["value"] + " ".join("%d:%.0f" % (c["chain_seed"], c["ms_per_step"]*1e3) for c in d["config"]["chains"]))
PY
  done
done
