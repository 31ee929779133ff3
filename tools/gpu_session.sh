python -m pytest tests -x -q -m gpu -k "hist or windows or ragged" 2>&1 | tail -3
HIST=20 python tools/bench_ragged.py 20000000 31 2>&1 | grep -v amdgpu.ids | head -12
