#!/bin/bash
mkdir -p gpurun_out/s6
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/s6/pytest.txt
tail -3 gpurun_out/s6/pytest.txt
for k in 33 41 47 51 55 63; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --sustain-steps 200 -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"
  KMX_LIB_VARIANT=r1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --sustain-steps 200 -k $k 2>/dev/null | python3 tools/bench_line.py "r1 k=$k"
done 2>&1 | tee gpurun_out/s6/k_sweep2.txt
