#!/bin/bash
# Inputs of the sharded chain's cost model, measured on ONE GPU: scan rounds per sweep against the band's
# width with what one rank reports per round (64 surprises) and with what EIGHT ranks report per round (a build with
# -DGG_SHARD_M=512: graphistician_amd/_build/variants/lib_m512.so), N=65536, K=64.
cd "$(dirname "$0")/../.."
out=gpurun_out/r4_shard_model.txt
one() {  # label lib band
  local j
  j=$(GG_LIB_PATH=$2 GG_SHARD_BAND=$3 timeout -k 5 120 python bench.py --shard --nodes 65536 --blocks 64 --steps 10 --warmup 6 2>/dev/null)
  echo "$1 band=$3 :: $(echo "$j" | python -c 'import json,sys; r=json.loads(sys.stdin.read()); c=r["config"]; print("%.1f sweeps/s  %.2f ms/sweep  rounds/sweep %.1f  us/round %.1f  changes/sweep %.0f  digest %d" % (r["value"], r["ms_per_step"], c["scan_rounds_per_sweep"], c["us_per_scan_round"], c["label_changes_in_timed_region"]/r["steps"], c["labels_digest"]))')" >> $out
  tail -1 $out
}
if [ -z "$ONLY_M512" ]; then for b in 4096 16384 65536 0; do one "M=64 " "" $b; done; fi
V=$PWD/graphistician_amd/_build/variants/lib_m512.so
for b in 4096 16384 32768 65536 0; do one "M=512" $V $b; done
