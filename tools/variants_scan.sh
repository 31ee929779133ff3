#!/bin/bash
# dev tool: build libkmx with extra -D flags and time the word-domain kernel paths (k=27 uniform reduce, ragged reduce, histogram b=12)
for v in "$@"; do
  python -c "from kmers_amd import build; build.build(force=True, extra='$v'.split())" >/dev/null 2>&1
  echo "== [$v]"
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline -k 27 2>/dev/null | python tools/bench_line.py "k=27 uniform"
  timeout 300 python tools/bench_ragged.py 20000000 2>/dev/null | head -1
  timeout 300 python tools/bench_hist.py 20000000 12,20 2>/dev/null
done
