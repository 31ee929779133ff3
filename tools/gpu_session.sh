#!/bin/bash
mkdir -p gpurun_out/s9
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s9/pytest_full.txt 2>&1
tail -5 gpurun_out/s9/pytest_full.txt
for spec in "1000 15000000" "10000 1500000" "300 50000000"; do set -- $spec
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --sustain-steps 0 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"
  KMX_LIB_VARIANT=r1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --sustain-steps 0 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "r1 L=$1"
done 2>&1 | tee gpurun_out/s9/long.txt
