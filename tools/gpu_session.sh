# scratch: the command list of the last gpurun call (round 3, run 5: all three changes on by default -- parity + sweeps)
mkdir -p gpurun_out/r3e
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3e/pytest.txt
tools/variants.sh default nobank > gpurun_out/r3e/variants.txt 2>&1
for k in 13 21 27 33 47 63; do python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 200 -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done > gpurun_out/r3e/k_sweep.txt
for spec in "100 150000000" "170 88000000" "200 75000000" "250 60000000" "256 58000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > gpurun_out/r3e/len_sweep.txt
