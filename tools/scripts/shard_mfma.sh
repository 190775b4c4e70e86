#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python -m pytest tests/test_sharded.py -x -q -m gpu > gpurun_out/shard_mfma_tests.log 2>&1 || { tail -30 gpurun_out/shard_mfma_tests.log; exit 1; }
tail -1 gpurun_out/shard_mfma_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_at_size.py -x -q -m gpu -k "shard" > gpurun_out/shard_mfma_at_size.log 2>&1 || { tail -30 gpurun_out/shard_mfma_at_size.log; exit 1; }
tail -1 gpurun_out/shard_mfma_at_size.log
for b in 0 65536 16384; do
j=$(GG_SHARD_BAND=$b timeout -k 5 120 python bench.py --shard --nodes 65536 --blocks 64 --steps 10 --warmup 6 2>/dev/null)
echo "band=$b :: $(echo "$j" | python -c 'import json,sys; r=json.loads(sys.stdin.read()); c=r["config"]; print("%.1f sweeps/s  %.2f ms/sweep  rounds/sweep %.1f  us/round %.1f  changes/sweep %.0f  digest %d" % (r["value"], r["ms_per_step"], c["scan_rounds_per_sweep"], c["us_per_scan_round"], c["label_changes_in_timed_region"]/r["steps"], c["labels_digest"]))')"
done
