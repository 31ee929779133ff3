#!/bin/bash
# dev tool: k = 26..30 at a forced 4 waves/SIMD against the default
for v in "-DKMX_BS_WAVES_FORCE=4" "" "-DKMX_BS_WAVES_FORCE=4" ""; do
  python -c "from kmers_amd import build; build.build(force=True, extra='$v'.split())" >/dev/null 2>&1
  for k in 26 27 28 29 30; do
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline -k $k 2>/dev/null | python tools/bench_line.py "[$v],k=$k"
  done
done
