#!/bin/bash
# dev tool: materialise mode, line-aligned write-back on/off, waves per SIMD
for v in "1 3" "0 3" "1 2" "0 2" "1 3" "0 3"; do
  set -- $v
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_WIN_ALIGNED=$1', '-DKMX_WIN_WAVES=$2'])" >/dev/null 2>&1
  echo "aligned=$1 waves=$2"
  timeout 300 python tools/bench_windows.py 20000000 150 2>&1 | grep "canon only"
  timeout 300 python tools/bench_windows.py 20000000 158 2>&1 | grep "canon only" | head -1
done
