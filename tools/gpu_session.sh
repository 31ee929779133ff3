out=gpurun_out/r02e; mkdir -p $out
for spec in "100 150000000" "250 60000000" "300 50000000" "1000 15000000" "10000 1500000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > $out/len_sweep.txt
python3 tools/bench_ragged.py 100000000 31 > $out/ragged_bench.txt 2>/dev/null
python3 tools/bench_ragged.py 100000000 21 >> $out/ragged_bench.txt 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "same box, headline" > $out/headline_same_box.txt
