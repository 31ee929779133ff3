mkdir -p gpurun_out/r3fin
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r3fin/pytest2.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3fin/bench_default.json 2> gpurun_out/r3fin/bench_default.err
