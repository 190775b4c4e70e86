#!/bin/bash
# records of round 3, final state, for profiles/: GPU test log, the driver's bench command, kernel-trace statistics of the
# configurations, the sharded chain through RCCL with one rank, the reference's call pattern, the 100-block chain
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
echo "start $(date +%T)" > $O/progress.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_driver_cmd_steps20.json 2> $O/bench.err; echo "bench rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 300 python bench.py --no-cpu --no-extras > $O/r03_bench_n4096_k16_steps200.json 2>> $O/bench.err; echo "bench200 rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 300 python bench.py --shard --nodes 65536 --blocks 64 --steps 6 --warmup 3 --no-cpu > $O/r03_shard_n65536_1rank_rccl.json 2>> $O/bench.err; echo "shard rc=$? $(date +%T)" >> $O/progress.log
timeout -k 10 300 python bench.py --shard --nodes 8192 --blocks 16 --steps 20 --warmup 5 --no-cpu > $O/r03_shard_n8192_1rank_rccl.json 2>> $O/bench.err; echo "shard8k rc=$? $(date +%T)" >> $O/progress.log
( GG_CALL_DBG=1 timeout -k 10 200 python tools/call_pattern.py ) > $O/r03_call_pattern.txt 2>&1; echo "call rc=$? $(date +%T)" >> $O/progress.log
( GG_EPILOGUE=0 GG_ONE_RIDE=0 timeout -k 10 200 python tools/call_pattern.py | head -4 | sed 's/^/[GG_EPILOGUE=0 GG_ONE_RIDE=0] /' ) >> $O/r03_call_pattern.txt 2>&1
timeout -k 10 200 python tools/wide_probe.py 4096 100 400 > $O/r03_wide_n4096_c100.txt 2>&1; echo "wide rc=$? $(date +%T)" >> $O/progress.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_a /tmp/p_b /tmp/p_c /tmp/p_d
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_a -- python3 $R/bench.py --no-cpu --no-extras > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_a -name "*kernel_stats.csv" | head -1) $O/r03_bench_n4096_k16_seedmix_kernel_stats.csv
python3 $R/tools/trace_gaps.py $(find /tmp/p_a -name "*kernel_trace.csv" | head -1) k_sweep_q -350 -50 > $O/r03_bench_n4096_k16_timed_window.txt 2>&1
echo "prof a $(date +%T)" >> $O/progress.log
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b -- python3 $R/bench.py --config eigen --steps 20 --warmup 3 --no-cpu > $O/r03_bench_eigen_n4096_d8.json 2> /dev/null
cp $(find /tmp/p_b -name "*kernel_stats.csv" | head -1) $O/r03_eigen_n4096_d8_kernel_stats.csv
echo "prof b $(date +%T)" >> $O/progress.log
cd $R
timeout -k 10 300 bash tools/scripts/ktrace.sh shard65k 400 -- python3 bench.py --shard --nodes 65536 --blocks 64 --steps 4 --warmup 2 --no-cpu > /dev/null 2>&1; cp gpurun_out/kt_shard65k.txt $O/r03_shard_n65536_kernel_trace.txt
timeout -k 10 300 bash tools/scripts/timeline.sh onecall 8 -- python3 tools/one_call.py 40 ll > /dev/null 2>&1; cp gpurun_out/tl_onecall.txt $O/r03_call_pattern_timeline.txt
timeout -k 10 300 bash tools/scripts/wide_trace.sh > /dev/null 2>&1; cp gpurun_out/wide_trace.txt $O/r03_wide_n4096_c100_kernel_trace.txt
echo "traces $(date +%T)" >> $O/progress.log
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/r03_gpu_tests.log 2>&1; echo "tests rc=$? $(date +%T)" >> $O/progress.log
tail -2 $O/r03_gpu_tests.log; cat $O/progress.log
