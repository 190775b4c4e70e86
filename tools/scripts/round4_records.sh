#!/bin/bash
# records of round 4, final state, for profiles/: bench lines, kernel-trace statistics of the configurations, the sharded chain
# through RCCL with one rank (kernel trace), PMC traffic of the fused launch and of the plain pass past the Infinity Cache
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
echo "start $(date +%T)" > $O/progress.log
timeout -k 10 300 python bench.py --no-cpu --no-extras > $O/r04_bench_n4096_k16_steps200.json 2> $O/bench.err; echo "bench200 rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 300 python bench.py --shard --nodes 65536 --blocks 64 --steps 10 --warmup 6 --no-cpu > $O/r04_shard_n65536_1rank_rccl.json 2>> $O/bench.err; echo "shard rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 300 python bench.py --shard --nodes 8192 --blocks 16 --steps 20 --warmup 5 --no-cpu > $O/r04_shard_n8192_1rank_rccl.json 2>> $O/bench.err; echo "shard8k rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 300 python tools/host_loop.py > $O/r04_host_loop.txt 2>&1; echo "host loop rc=$? $(date +%T)" >> $O/progress.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_a /tmp/p_b /tmp/p_c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_a -- python3 $R/bench.py --no-cpu --no-extras --repeats 3 > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_a -name "*kernel_stats.csv" | head -1) $O/r04_bench_n4096_k16_seedmix_kernel_stats.csv
python3 $R/tools/trace_gaps.py $(find /tmp/p_a -name "*kernel_trace.csv" | head -1) k_sweep_q -350 -50 > $O/r04_bench_n4096_k16_timed_window.txt 2>&1
echo "prof a $(date +%T)" >> $O/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b -- python3 $R/bench.py --config eigen --steps 20 --warmup 3 --no-cpu > $O/r04_bench_eigen_n4096_d8.json 2> /dev/null
cp $(find /tmp/p_b -name "*kernel_stats.csv" | head -1) $O/r04_eigen_n4096_d8_kernel_stats.csv
echo "prof b $(date +%T)" >> $O/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c -- python3 $R/bench.py --nodes 65536 --blocks 64 --steps 20 --warmup 2 --no-cpu --no-extras --repeats 1 > $O/r04_bench_n65536_k64_one_gpu_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_c -name "*kernel_stats.csv" | head -1) $O/r04_n65536_k64_kernel_stats.csv
python3 $R/tools/trace_gaps.py $(find /tmp/p_c -name "*kernel_trace.csv" | head -1) 400 > $O/r04_n65536_k64_timed_window.txt 2>&1
echo "prof c $(date +%T)" >> $O/progress.log
cd $R
timeout -k 10 300 bash tools/scripts/ktrace.sh shard65k 300 -- python3 bench.py --shard --nodes 65536 --blocks 64 --steps 4 --warmup 4 --no-cpu > /dev/null 2>&1; cp gpurun_out/kt_shard65k.txt $O/r04_shard_n65536_kernel_trace.txt
echo "trace shard $(date +%T)" >> $O/progress.log
timeout -k 10 300 python bench.py --nodes 65536 --blocks 64 --steps 20 --warmup 2 --no-cpu --no-extras > $O/r04_bench_n65536_k64_one_gpu.json 2>> $O/bench.err; echo "bench 65536 rc=$? $(date +%T)" >> $O/progress.log
cat $O/progress.log
