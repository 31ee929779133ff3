#!/bin/bash
# scratch: the GPU session of the moment
timeout 300 python3 tools/step_times2.py 5 2>&1 | grep -v amdgpu | cut -c1-330 | head -9
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Average Graphics Package Power\|sclk clock" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/smi.txt &
python3 bench.py --gpus 1 --steps 2000 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 0 2>/dev/null | python3 tools/bench_line.py "[2000 steps]"
wait
sort gpurun_out/smi.txt | uniq -c | sort -rn | head -8
