# Build variants of gg_model.hip (the scan kernels' translation unit) with different compile-time settings, one library each
# under graphistician_amd/_build/variants/ (tuning only; the headers carry the setting that is kept):
#   bash tools/scripts/model_variants.sh name1 "flags1" name2 "flags2" ...      (run AFTER make: the other objects are reused)
set -e
cd "$(dirname "$0")/../../graphistician_amd/csrc"
mkdir -p ../_build/variants
B=../_build
pids=()
names=()
while [ $# -ge 2 ]; do
  N=$1; F=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $F -c gg_model.hip -o /tmp/model_$N.o 2> /tmp/model_$N.err \
    && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/variants/lib_$N.so $B/gg_data.o /tmp/model_$N.o $B/gg_sweepq.o $B/gg_eigen.o $B/gg_latent.o $B/gg_selftest.o $B/gg_hash.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib \
    && echo "$N: built ($F)" || { echo "$N: failed"; tail -3 /tmp/model_$N.err; } ) &
done
wait
