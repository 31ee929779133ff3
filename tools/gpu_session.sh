mkdir -p gpurun_out/r3u
for b in 1 2 4 8; do echo "== blocks per CU $b"; KMX_ROLL_BPC=$b python3 tools/bench_dirty.py 2>&1 | grep -v amdgpu | head -5; done > gpurun_out/r3u/dirty_bpc.txt
