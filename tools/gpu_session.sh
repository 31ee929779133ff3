mkdir -p gpurun_out/r3fin
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3fin/bench_default2.json 2> gpurun_out/r3fin/bench_default2.err
python bench.py --config 2 -k 63 --no-traffic --no-cpu-baseline > gpurun_out/r3fin/bench_k63.json 2>/dev/null
python bench.py --config 4 --no-traffic --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 20 > gpurun_out/r3fin/bench_hist20.json 2>/dev/null
