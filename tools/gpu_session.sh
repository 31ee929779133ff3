python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for spec in "100 150000000" "112 130000000" "75 200000000" "50 300000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done
python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len 100 --reads-per-gpu 150000000 -k 21 2>/dev/null | python3 tools/bench_line.py "L=100 k=21"
