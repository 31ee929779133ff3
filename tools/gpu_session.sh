#!/bin/bash
mkdir -p gpurun_out/s19
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s19/pytest.txt 2>&1; tail -2 gpurun_out/s19/pytest.txt
for spec in "250 60000000" "220 68000000" "200 75000000" "170 88000000" "161 93000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"
done | tee gpurun_out/s19/len.txt
