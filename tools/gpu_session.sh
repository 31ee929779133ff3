# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4t; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for spec in "63 200 75000000" "63 250 60000000" "33 250 60000000" "47 208 72000000" "63 150 100000000"; do set -- $spec
  $B -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "k=$1 L=$2"; done > $O/k2_long.txt 2>&1; cat $O/k2_long.txt
