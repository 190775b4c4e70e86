# kernel timeline of one overlapped eigenmodel sweep: bash tools/scripts/eig_trace.sh NAME   (env passes through)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$1
( cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$1 -- python3 bench.py --config eigen --steps 3 --warmup 2 --no-cpu > /dev/null 2> $R/gpurun_out/eig_trace_$1.err )
python3 $R/tools/eig_timeline.py $(find /tmp/kt_$1 -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/eig_trace_$1.txt
cat $R/gpurun_out/eig_trace_$1.txt
