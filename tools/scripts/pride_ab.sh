#!/bin/bash
# mixing chains with and without the pass riding with the parameter step (GG_PRIDE): sweeps/s per chain seed, N=4096, K=16
cd "$(dirname "$0")/../.."
for p in 0 1; do
  for s in 2 3 4 1; do
    GG_PRIDE=$p GG_BENCH_CHAIN_SEED=$s python bench.py --no-cpu --no-extras --steps 200 --warmup 20 --repeats 5 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); c=r['config']['chains'][0]
print('GG_PRIDE=$p seed $s: %.0f sweeps/s (%.1f us) changes/sweep %.1f rounds/sweep %.1f forms %s' % (c['value'], c['ms_per_step']*1e3, c['label_changes_per_sweep'], c['scan_rounds_per_sweep'], c['weight_model_sweep_forms']))"
  done
done
