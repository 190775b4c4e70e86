# HBM traffic of the fused launch (N=4096, K=16) and of the plain weight pass at N=32768 (past the Infinity Cache), round 3:
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, --kernel-trace only, program directly after --
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_f1 /tmp/p_w1 /tmp/p_f2 /tmp/p_w2
export GG_BENCH_CHAIN_SEED=1
( cd $R && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f1 -- python3 bench.py --no-cpu --no-extras --steps 60 > /dev/null 2>&1 )
( cd $R && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w1 -- python3 bench.py --no-cpu --no-extras --steps 60 > /dev/null 2>&1 )
python3 $R/tools/pmc_traffic.py $(ls /tmp/p_f1/*/*counter_collection.csv | head -1) $(ls /tmp/p_w1/*/*counter_collection.csv | head -1) $O/r04_pmc_traffic_n4096.json "k_sweep_q<3=134217728" "k_sweep_q<2=4194304" > /dev/null
cp $(ls /tmp/p_f1/*/*counter_collection.csv | head -1) $O/r04_pmc_fetch_size_n4096.csv
cp $(ls /tmp/p_w1/*/*counter_collection.csv | head -1) $O/r04_pmc_write_size_n4096.csv
unset GG_BENCH_CHAIN_SEED
( cd $R && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f2 -- python3 tools/bulk_only.py 32768 64 3 > /dev/null 2>&1 )
( cd $R && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w2 -- python3 tools/bulk_only.py 32768 64 3 > /dev/null 2>&1 )
python3 $R/tools/pmc_traffic.py $(ls /tmp/p_f2/*/*counter_collection.csv | head -1) $(ls /tmp/p_w2/*/*counter_collection.csv | head -1) $O/r04_pmc_traffic_n32768.json "k_bulk_w=8589934592" "k_bulk_adj=268435456" > /dev/null
cat $O/r04_pmc_traffic_n4096.json $O/r04_pmc_traffic_n32768.json | grep -v "kernels\|\[\|\]" | head -40
ls -la $O
