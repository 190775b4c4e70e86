# long-run parity of the round's final state: engine vs the multi-threaded CPU oracle, labels and parameters bit for bit
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_soak.txt
echo "# tests/soak_parity.py N K sweeps (engine vs oracle_fast, compared along the way), final state of round 3" > $O
for cfg in "512 8 20000" "4096 16 400" "1024 40 600" "1500 64 100" "700 12 2000"; do
  timeout -k 10 300 python tests/soak_parity.py $cfg 2>&1 | tail -1 >> $O || echo "FAILED: $cfg" >> $O
done
echo "# GG_SCAN=1 (resident grid with hypothesis rounds from a random start)" >> $O
for cfg in "4096 16 30" "2048 32 40" "1500 64 100"; do
  GG_SCAN=1 timeout -k 10 300 python tests/soak_parity.py $cfg 2>&1 | tail -1 >> $O || echo "FAILED GG_SCAN=1: $cfg" >> $O
done
echo "# GG_SCAN=1 GG_SCAN_LCAP=7 (change list of 7 entries: acceptances cut short, chip-wide flushes)" >> $O
GG_SCAN=1 GG_SCAN_LCAP=7 timeout -k 10 300 python tests/soak_parity.py 2048 32 30 2>&1 | tail -1 >> $O || echo "FAILED LCAP" >> $O
i=21
for env in "GG_X=1" "GG_SCAN=1" "GG_SCAN=0" "GG_SCAN_SPIN=2" "GG_AHEAD=0" "GG_SCAN=1 GG_SCAN_LCAP=3" "GG_EPILOGUE=0" "GG_ONE_RIDE=0" "GG_SCAN=0 GG_AHEAD=0"; do
  echo "fuzz $env FUZZ_EXTENDED=1 300 cases seed $i: $(env $env FUZZ_EXTENDED=1 timeout -k 10 600 python tests/fuzz_parity.py 300 $i 2>&1 | tail -1)" >> $O
  i=$((i+1))
done
cat $O
echo "float-model shape sweep (tests/fuzz_float_models.py): $(timeout -k 10 900 python tests/fuzz_float_models.py 2>&1 | tail -1)" >> $O
tail -2 $O
echo "# the call pattern (one sweep per call + log-likelihood: the epilogue form), 3 shapes x 300 iterations against the oracle" >> $O
timeout -k 10 600 python tests/soak_calls.py >> $O 2>&1 || echo "FAILED: soak_calls" >> $O
echo "# larger networks and more than 64 blocks" >> $O
for cfg in "8192 16 300" "16384 32 60" "8192 64 60" "2048 100 40" "3000 128 20" "1500 256 20"; do
  timeout -k 10 500 python tests/soak_parity.py $cfg 2>&1 | tail -1 >> $O || echo "FAILED: $cfg" >> $O
done
tail -8 $O
