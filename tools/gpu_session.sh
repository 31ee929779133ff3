mkdir -p gpurun_out/r3sh
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "five_word or frame or short" 2>&1 | tail -3 > gpurun_out/r3sh/pytest.txt
for spec in "36 400000000" "50 300000000" "62 240000000" "75 200000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > gpurun_out/r3sh/len_sweep.txt
for spec in "36 400000000" "50 300000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 -k 21 2>/dev/null | python3 tools/bench_line.py "k=21 L=$1"; done >> gpurun_out/r3sh/len_sweep.txt
