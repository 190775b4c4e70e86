# Build variants of gg_sweepq.hip (the fused kernel's own translation unit) with different compiler settings, one library each
# under graphistician_amd/_build/variants/ (tuning only; the Makefile carries the setting that is kept):
#   bash tools/scripts/sweepq_variants.sh name1 "flags1" name2 "flags2" ...
set -e
cd "$(dirname "$0")/../../graphistician_amd/csrc"
mkdir -p ../_build/variants
B=../_build
while [ $# -ge 2 ]; do
  N=$1; F=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $F -c gg_sweepq.hip -o /tmp/sweepq_$N.o 2> /tmp/sweepq_$N.err || { echo "$N: compile failed"; tail -3 /tmp/sweepq_$N.err; continue; }
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/variants/lib_$N.so $B/gg_data.o $B/gg_model.o /tmp/sweepq_$N.o $B/gg_eigen.o $B/gg_latent.o $B/gg_selftest.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo "$N: built ($F)"
done
