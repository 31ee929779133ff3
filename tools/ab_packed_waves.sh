#!/bin/bash
# dev tool: packed-input bit-sliced kernel at 3 vs 4 waves/SIMD
for v in 4 3 4 3; do
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_BSP_WAVES=$v'])" >/dev/null 2>&1
  for k in 31 21 15; do
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --packed -k $k 2>/dev/null | python tools/bench_line.py packed,waves=$v,k=$k
  done
done
