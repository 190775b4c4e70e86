cd $GRAFT_REPO_ROOT/graphistician_amd/_build
cp libgraphgibbs.so lib_orig.so
for v in orig gg_uj16 gg_uj8; do
  if [ $v = orig ]; then cp lib_orig.so libgraphgibbs.so; else cp lib$v.so libgraphgibbs.so; fi
  cd $GRAFT_REPO_ROOT
  GG_ROUND_DBG=1 python bench.py --nodes 65536 --blocks 64 --steps 6 --warmup 2 --burnin 12 --no-cpu --no-extras > gpurun_out/ab.json 2> gpurun_out/ab.err
  echo "== $v"; grep "scan dbg" gpurun_out/ab.err | tail -2 | cut -c1-150
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('big', d['value'], d['ms_per_step'])"
  cd $GRAFT_REPO_ROOT/graphistician_amd/_build
done
cp lib_orig.so libgraphgibbs.so
