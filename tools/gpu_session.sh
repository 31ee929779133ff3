mkdir -p gpurun_out/r3wu
timeout 1500 python -m pytest tests/ -x -q -m gpu -k "windows or materialise or fuzz" 2>&1 | tail -3 > gpurun_out/r3wu/pytest8.txt
timeout 600 python tools/bench_windows.py 2>&1 | grep -v amdgpu > gpurun_out/r3wu/windows_bench8.txt
