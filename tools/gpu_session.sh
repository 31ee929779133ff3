python -m pytest tests -x -q -m gpu -k "fastx" 2>&1 | tail -2; python tools/bench_fastx.py 2>/dev/null
