mkdir -p gpurun_out/r3nt
timeout 600 python tools/bench_windows2.py 2>&1 | grep -v amdgpu > gpurun_out/r3nt/windows2_bench.txt
timeout 300 python tools/bench_fastx.py 2>&1 | grep -v amdgpu > gpurun_out/r3nt/fastx_bench.txt
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3nt/pytest.txt
