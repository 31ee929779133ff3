python -m pytest tests -x -q -m gpu -k "uniform_lengths or aligned or dirty or blanked" 2>&1 | tail -2
for spec in "75 200000000" "50 300000000" "64 230000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done
python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len 50 --reads-per-gpu 300000000 -k 21 2>/dev/null | python3 tools/bench_line.py "L=50 k=21"
