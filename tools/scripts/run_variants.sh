# quiet-chain probe for every library under graphistician_amd/_build/variants/ (on the GPU box)
for f in graphistician_amd/_build/variants/lib_*.so; do
  n=$(basename $f .so)
  r=$(GG_LIB_PATH=$PWD/$f timeout -k 10 100 python tools/quiet_probe.py both 2>&1 | tail -1 | sed 's/.*: //')
  w=$(GG_LIB_PATH=$PWD/$f timeout -k 10 100 python tools/quiet_probe.py w 2>&1 | tail -1 | sed 's/.*: //')
  echo "$n  both $r   w $w"
done
