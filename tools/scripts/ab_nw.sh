cd $GRAFT_REPO_ROOT/graphistician_amd/_build
cp libgraphgibbs.so lib_orig.so
for v in orig gg_nw2 gg_nw1 orig; do
  if [ $v = orig ]; then cp lib_orig.so libgraphgibbs.so; else cp lib$v.so libgraphgibbs.so; fi
  cd $GRAFT_REPO_ROOT
  echo "== $v"
  for seed in 2 4; do GG_BENCH_CHAIN_SEED=$seed python bench.py --no-cpu --no-extras --steps 200 > gpurun_out/ab.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('seed $seed', round(d['value']), round(d['ms_per_step']*1000,1))"; done
  cd $GRAFT_REPO_ROOT/graphistician_amd/_build
done
cp lib_orig.so libgraphgibbs.so
