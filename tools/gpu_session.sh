# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4y; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python3 tools/_exp_long2.py 2>/dev/null | tail -2
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for spec in "300 50000000" "1000 15000000" "10000 1500000"; do set -- $spec
  $B --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done | tee $O/long.txt
python3 tools/bench_ragged.py 100000000 31 2>/dev/null | tee $O/ragged.txt
python3 tools/bench_long_ragged.py 6e9 31 2>/dev/null | tee $O/long_ragged.txt
