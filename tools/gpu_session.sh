python -m pytest tests -x -q -m gpu -k "hist or fuzz" 2>&1 | tail -2
python3 tools/bench_hist.py 100000000 20,21,22,23 2>/dev/null
