#!/bin/bash
# scratch: the GPU session of the moment
for i in 1 2 3; do for v in default w2; do
python3 tools/bench_variant.py $v --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 300 2>/dev/null | python3 tools/bench_line.py "[$v]"; done; done
