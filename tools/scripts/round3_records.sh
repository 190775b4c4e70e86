#!/bin/bash
# records of round 3 for profiles/: GPU test log, the driver's bench command, kernel-trace statistics of the three configs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
echo "tests $(date +%T)" > $O/progress.log
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/r03_gpu_tests.log 2>&1; echo "tests rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_driver_cmd_steps20.json 2> $O/bench.err; echo "bench rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu --no-extras > $O/r03_bench_n4096_k16_steps200.json 2>> $O/bench.err; echo "bench200 rc=$? $(date +%T)" >> $O/progress.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_a /tmp/p_b /tmp/p_c
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_a -- python3 $R/bench.py --no-cpu --no-extras > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_a -name "*kernel_stats.csv" | head -1) $O/r03_bench_n4096_k16_seedmix_kernel_stats.csv
python3 $R/tools/trace_gaps.py $(find /tmp/p_a -name "*kernel_trace.csv" | head -1) k_sweep_q -350 -50 > $O/r03_bench_n4096_k16_timed_window.txt 2>&1
echo "prof a $(date +%T)" >> $O/progress.log
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b -- python3 $R/bench.py --config eigen --steps 20 --warmup 3 --no-cpu > $O/bench_eigen_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_b -name "*kernel_stats.csv" | head -1) $O/r03_eigen_n4096_d8_kernel_stats.csv
echo "prof b $(date +%T)" >> $O/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c -- python3 $R/bench.py --nodes 65536 --blocks 64 --steps 20 --warmup 2 --no-cpu --no-extras > $O/bench_n65536_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_c -name "*kernel_stats.csv" | head -1) $O/r03_n65536_k64_kernel_stats.csv
python3 $R/tools/trace_gaps.py $(find /tmp/p_c -name "*kernel_trace.csv" | head -1) "k_bulk_w" -60 -1 > $O/r03_n65536_k64_timed_window.txt 2>&1
echo "prof c $(date +%T)" >> $O/progress.log
tail -2 $O/r03_gpu_tests.log; cat $O/progress.log
