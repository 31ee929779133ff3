#!/bin/bash
# scratch: the GPU session of the moment
for a in "76 200000000" "101 150000000" "151 100000000" "251 60000000" "301 50000000"; do set -- $a
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done
for a in "21 151" "25 151" "27 101" "21 101" "33 151" "51 151"; do set -- $a
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 -k $1 --read-len $2 2>/dev/null | python3 tools/bench_line.py "k=$1 L=$2"; done
