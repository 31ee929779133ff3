#!/bin/bash
# scratch: the GPU session of the moment
for v in "" hg8 hg32 hgad "" hg8; do echo "== '$v'"
KMX_DEV_LIB=$v timeout 300 python3 tools/bench_small_hist.py 2>&1 | grep "n ="
python3 tools/bench_variant.py ${v:-default} --config 4 --no-cpu-baseline --no-traffic --steps 5 --warmup 2 --sustain-steps 0 2>/dev/null | python3 tools/bench_line.py "[hist20]"
done
