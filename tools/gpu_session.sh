bash tools/profile_round.sh gpurun_out/r03c > gpurun_out/r03c.log 2>&1
python3 tools/bench_elem.py 2>/dev/null | grep -v amdgpu > gpurun_out/r03c/elem_bench.txt
