mkdir -p gpurun_out/r3st
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3st/bench_default.json 2> gpurun_out/r3st/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --settle-steps 0 --no-cpu-baseline --no-traffic > gpurun_out/r3st/bench_settle0.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/r3st/bench_settle30.json 2>/dev/null
