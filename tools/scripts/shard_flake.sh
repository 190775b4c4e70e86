#!/bin/bash
# diagnose an intermittent mismatch of the sharded chain: the same case several times, the oracle beside it sweep by sweep
cd "$(dirname "$0")/../.."
run() {  # world N K form band
  echo "== world=$1 N=$2 K=$3 form=$4 band=$5 fold1=$GG_SHARD_FOLD1"
  GG_SHARD_DIAG=1 GG_SHARD_BAND=$5 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) tests/shard_worker.py $2 $3 3 1 1 $4 2>&1 | grep -E "DIAG|Error" | grep -v "differ at \[\], p True, m True | w labels differ at \[\], mu True, m True" | head -8
}
export GG_SHARD_FOLD1=0
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do run 3 9000 12 blocks 65536; done
for i in 1 2; do run 3 9000 12 range 65536; done
