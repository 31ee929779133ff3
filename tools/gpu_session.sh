mkdir -p gpurun_out/r3dr
timeout 600 python tools/bench_dirty.py > gpurun_out/r3dr/dirty.txt 2>&1
HIST=20 timeout 600 python tools/bench_dirty.py > gpurun_out/r3dr/dirty_hist.txt 2>&1
