set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
# 1. kernel trace + stats of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --no-cpu --no-extras > $O/bench_under_rocprof.json 2> /dev/null
cp $(ls /tmp/p_stats/*/*kernel_stats.csv | head -1) $O/r02_bench_n4096_k16_kernel_stats.csv
python3 $R/tools/trace_gaps.py $(ls /tmp/p_stats/*/*kernel_trace.csv | head -1) k_sweep_q -350 -50 > $O/r02_bench_n4096_k16_timed_window.txt
# 2. PMC traffic, separate passes, fused sweep at N=4096 and bulk kernels at N=32768
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f1 -- python3 $R/bench.py --no-cpu --no-extras --steps 60 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w1 -- python3 $R/bench.py --no-cpu --no-extras --steps 60 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $(ls /tmp/p_f1/*/*counter_collection.csv | head -1) $(ls /tmp/p_w1/*/*counter_collection.csv | head -1) $O/r02_pmc_traffic_n4096.json "k_sweep_q<3=134217728" "k_sweep_q<2=4194304" > /dev/null
cp $(ls /tmp/p_f1/*/*counter_collection.csv | head -1) $O/r02_pmc_fetch_size_n4096.csv
cp $(ls /tmp/p_w1/*/*counter_collection.csv | head -1) $O/r02_pmc_write_size_n4096.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f2 -- python3 $R/tools/bulk_only.py 32768 64 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w2 -- python3 $R/tools/bulk_only.py 32768 64 3 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $(ls /tmp/p_f2/*/*counter_collection.csv | head -1) $(ls /tmp/p_w2/*/*counter_collection.csv | head -1) $O/r02_pmc_traffic_n32768.json "k_bulk_w=8589934592" "k_bulk_adj=268435456" > /dev/null
cp $(ls /tmp/p_f2/*/*counter_collection.csv | head -1) $O/r02_pmc_fetch_size_bulk_n32768_k64.csv
cp $(ls /tmp/p_w2/*/*counter_collection.csv | head -1) $O/r02_pmc_write_size_bulk_n32768_k64.csv
cat $O/r02_pmc_traffic_n4096.json $O/r02_pmc_traffic_n32768.json
cat $O/r02_bench_n4096_k16_timed_window.txt
# 3. a chain that keeps changing labels (bench chain seed 2: ~60 changes per sweep): kernel stats of the resident-scan form
GG_BENCH_CHAIN_SEED=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mix -- python3 $R/bench.py --no-cpu --no-extras > $O/bench_mix_under_rocprof.json 2> /dev/null
cp $(ls /tmp/p_mix/*/*kernel_stats.csv | head -1) $O/r02_mix_seed2_kernel_stats.csv
head -8 $O/r02_mix_seed2_kernel_stats.csv
