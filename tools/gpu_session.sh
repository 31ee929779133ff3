#!/bin/bash
mkdir -p gpurun_out/s12
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s12/pytest_full.txt 2>&1
tail -4 gpurun_out/s12/pytest_full.txt
for spec in "250 60000000" "200 75000000" "170 88000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"
  KMX_LIB_VARIANT=r1 python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "r1 L=$1"
done
