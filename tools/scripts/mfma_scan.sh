#!/bin/bash
# The resident scan's evaluation on the matrix cores (gg_lp_mfma): parity on the 33..64-block paths, then configs[3] on one GPU
# with the scan's per-round timers.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity_sbm.py tests/test_gpu_edge_cases.py -x -q -m gpu > $O/mfma_parity.log 2>&1 || { tail -30 $O/mfma_parity.log; exit 1; }
tail -2 $O/mfma_parity.log
timeout -k 10 400 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -k "not eigen and not latent" > $O/mfma_at_size.log 2>&1 || { tail -30 $O/mfma_at_size.log; exit 1; }
tail -2 $O/mfma_at_size.log
GG_ROUND_DBG=1 timeout -k 10 300 python bench.py --nodes 65536 --blocks 64 --steps 20 --warmup 2 --burnin 12 --no-cpu --no-extras --repeats 1 > $O/mfma_big.json 2> $O/mfma_big.err || { tail -20 $O/mfma_big.err; exit 1; }
grep "scan dbg" $O/mfma_big.err | tail -4
python - <<PY
import json
d = json.loads([l for l in open("$O/mfma_big.json") if l.startswith("{")][-1])
print("N=65536 K=64: %.1f sweeps/s, %.3f ms per sweep; changes/sweep %.1f" % (d["value"], d["ms_per_step"], d["config"].get("label_changes_per_sweep", -1)))
PY
