#!/bin/bash
# dev tool: the dirty-read list on / off (headline on clean input, and with N-holding reads)
for v in 1 0 1 0; do
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_BS_DIRTY=$v'])" >/dev/null 2>&1
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py dirty_list=$v
  timeout 300 python tools/bench_dirty.py 2>&1 | tail -5 | head -3
done
