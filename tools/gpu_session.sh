out=gpurun_out/r02f; mkdir -p $out
python3 bench.py --config 4 --no-traffic --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 20 > $out/bench_hist20.json 2> /dev/null
python3 bench.py --config 4 --dist-single --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 0 > $out/bench_hist20_rccl1.json 2> /dev/null
python3 tools/bench_hist.py 100000000 12,16,20,22 > $out/hist_bench.txt 2>/dev/null
