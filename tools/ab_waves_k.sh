#!/bin/bash
# dev tool: 3 vs 4 waves/SIMD of the ASCII bit-sliced kernel for several k (fewer counters at small k: 4 waves fit without spills up to k = 22)
for v in 4 3 4 3; do
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_BS_WAVES=$v'])" >/dev/null 2>&1
  for k in 15 19 21 23 25 27; do
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline -k $k 2>/dev/null | python tools/bench_line.py waves=$v,k=$k
  done
done
