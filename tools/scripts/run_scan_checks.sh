set -e
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_edge_cases.py -x -q -m gpu -k "resident or adaptive or timeout" > gpurun_out/scan1.log 2>&1 || { tail -30 gpurun_out/scan1.log; exit 1; }
tail -3 gpurun_out/scan1.log
GG_SCAN=1 timeout -k 10 300 python tests/fuzz_parity.py 40 7 > gpurun_out/fuzz_scan1.log 2>&1 || { tail -20 gpurun_out/fuzz_scan1.log; exit 1; }
tail -2 gpurun_out/fuzz_scan1.log
GG_SCAN=1 GG_SCAN_LCAP=3 timeout -k 10 300 python tests/fuzz_parity.py 25 11 > gpurun_out/fuzz_scan2.log 2>&1 || { tail -20 gpurun_out/fuzz_scan2.log; exit 1; }
tail -2 gpurun_out/fuzz_scan2.log
