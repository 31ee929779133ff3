#!/bin/bash
mkdir -p gpurun_out/s10
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>gpurun_out/s10/e1.txt | tee gpurun_out/s10/b_default.json | python3 tools/bench_line.py "default"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "default again"
python3 bench.py --config 3 --no-cpu-baseline --no-traffic --sustain-steps 100 2>gpurun_out/s10/e3.txt | tee gpurun_out/s10/b_c3.json | python3 tools/bench_line.py "config3"
python3 bench.py --config 4 --no-cpu-baseline --no-traffic --sustain-steps 20 --steps 5 --warmup 2 2>gpurun_out/s10/e4.txt | tee gpurun_out/s10/b_c4.json | cut -c1-1800
python3 bench.py --config 4 --dist-single --no-cpu-baseline --no-traffic --sustain-steps 0 --steps 3 --warmup 1 2>gpurun_out/s10/e4d.txt | tee gpurun_out/s10/b_c4d.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 dist-single', d['histogram'], d['rccl_ranks'], d['parity_vs_oracle'])"
python3 bench.py --config 2 -k 63 --no-traffic --sustain-steps 100 2>gpurun_out/s10/e63.txt | tee gpurun_out/s10/b_k63.json | python3 tools/bench_line.py "k63"
python3 bench.py --packed --no-cpu-baseline --no-traffic --sustain-steps 100 2>/dev/null | python3 tools/bench_line.py "packed"
tail -3 gpurun_out/s10/e4.txt gpurun_out/s10/e4d.txt gpurun_out/s10/e3.txt
