mkdir -p gpurun_out/r3hh
python tools/hist_host_time.py > gpurun_out/r3hh/host2.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "histogram" 2>&1 | tail -3 > gpurun_out/r3hh/pytest.txt
python3 bench.py --config 4 --no-traffic --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 20 > gpurun_out/r3hh/bench_hist20.json 2> /dev/null
