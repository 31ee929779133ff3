python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 400 2>/dev/null | python tools/bench_line.py default
python tools/bench_dirty.py 2>/dev/null
