#!/bin/bash
# dev tool: the 16-word frame (161..256 bp reads) at 2 vs 3 waves/SIMD
for v in 3 2 3 2; do
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_BS_WAVES16=$v'])" >/dev/null 2>&1
  for k in 31 21; do
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --read-len 250 --reads-per-gpu 60000000 -k $k 2>/dev/null | python tools/bench_line.py waves16=$v,L=250,k=$k
  done
done
