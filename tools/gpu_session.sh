python -m pytest tests -x -q -m gpu -k "long or segment" 2>&1 | tail -2
for spec in "300 50000000" "400 37000000" "257 58000000" "1000 15000000" "10000 1500000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done
