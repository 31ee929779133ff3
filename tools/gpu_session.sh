#!/bin/bash
# scratch: the GPU session of the moment
for i in 1 2 3 4 5 6; do for v in default qfea9; do
python3 tools/bench_variant.py $v --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 500 2>/dev/null | python3 tools/bench_line.py "[$v]"; done; done
for v in default qfea9; do python3 tools/bench_variant.py $v --config 2 -k 63 --no-cpu-baseline --no-traffic --sustain-steps 300 2>/dev/null | python3 tools/bench_line.py "[k63 $v]"; done
for v in default qfea9; do python3 tools/bench_variant.py $v --config 2 -k 63 --no-cpu-baseline --no-traffic --sustain-steps 300 2>/dev/null | python3 tools/bench_line.py "[k63 $v]"; done
