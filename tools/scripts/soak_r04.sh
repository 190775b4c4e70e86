# long-run parity of the round's final state: engine vs the multi-threaded CPU oracle, labels and parameters bit for bit
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_soak.txt
echo "# tests/soak_parity.py N K sweeps (engine vs oracle_fast, compared along the way), final state of round 4" > $O
for cfg in "512 8 8000" "4096 16 400" "1024 40 400" "1500 64 100" "1200 50 200" "900 37 200" "8192 16 200" "2048 100 40" "1500 256 20"; do
  timeout -k 10 300 python tests/soak_parity.py $cfg 2>&1 | tail -1 >> $O || echo "FAILED: $cfg" >> $O
  tail -1 $O
done
echo "# the pass riding with the parameter step (GG_PRIDE=1, off by default) and the scan's change list applied by k_apply_moves" >> $O
for cfg in "4096 16 300" "2048 32 100"; do
  GG_PRIDE=1 GG_SCAN=1 timeout -k 10 300 python tests/soak_parity.py $cfg 2>&1 | tail -1 >> $O || echo "FAILED GG_PRIDE=1: $cfg" >> $O
  tail -1 $O
done
i=41
for env in "GG_X=1" "GG_SCAN=1" "GG_SCAN=0" "GG_AHEAD=0" "GG_SCAN=1 GG_SCAN_LCAP=3" "GG_PRIDE=1 GG_SCAN=1" "GG_PRIDE=1 GG_SCAN=1 GG_SCAN_LCAP=4" "GG_EPILOGUE=0" "GG_MFMA_SMALL=1 GG_SCAN=1"; do
  echo "fuzz $env FUZZ_EXTENDED=1 200 cases seed $i: $(env $env FUZZ_EXTENDED=1 timeout -k 10 600 python tests/fuzz_parity.py 200 $i 2>&1 | tail -1)" >> $O
  tail -1 $O
  i=$((i+1))
done
echo "float-model shape sweep (tests/fuzz_float_models.py): $(timeout -k 10 900 python tests/fuzz_float_models.py 2>&1 | tail -1)" >> $O
tail -1 $O
echo "# the call pattern (one sweep per call + log-likelihood: the epilogue form), 3 shapes x 300 iterations against the oracle" >> $O
timeout -k 10 600 python tests/soak_calls.py >> $O 2>&1 || echo "FAILED: soak_calls" >> $O
tail -3 $O
