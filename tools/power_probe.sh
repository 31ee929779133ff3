#!/bin/bash
# dev tool: package power and shader clock (rocm-smi) while bench.py loops on one build variant
# usage: tools/power_probe.sh "<extra -D flags>" ...
for v in "$@"; do
  python -c "from kmers_amd import build; build.build(force=True, extra='$v'.split())" >/dev/null 2>&1
  (timeout 60 python bench.py --steps 2500 --warmup 3 --no-cpu-baseline > /tmp/pp.log 2>&1 &)
  sleep 9
  s=""
  for i in 1 2 3 4; do s="$s $(rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Package Power \(W\)" | sed 's/.*(\([0-9]*\)Mhz.*/\1MHz/; s/.*: \([0-9.]*\)$/\1W/' | tr '\n' ' ')"; sleep 1; done
  wait; sleep 6
  echo "[$v] $s | $(python tools/bench_line.py x < /tmp/pp.log 2>/dev/null | cut -c1-80)"
done
