#!/bin/bash
# dev tool: time the bit-sliced kernel with phases compiled out (results are wrong by design)
for a in "$@"; do
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_BS_ABLATE=$a'])" >/dev/null 2>&1
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py ablate=$a
done
