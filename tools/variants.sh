#!/bin/bash
# dev tool: build libkmx with extra -D flags and bench each variant (same box, same run)
for v in "$@"; do
  python -c "from kmers_amd import build; build.build(force=True, extra='$v'.split())" >/dev/null 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "[$v]"
done
