# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4k; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for v in default k2w3; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  echo "== $v"
  for k in 33 41 47 55 63; do $B -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done
done > $O/k2.txt 2>&1; cat $O/k2.txt
unset KMX_LIB_VARIANT
for k in 13 17 21 25 27 29 31; do $B -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done > $O/ksweep.txt; cat $O/ksweep.txt
$B --hash 2>/dev/null | python3 tools/bench_line.py "k=31 hash" | tee $O/hash.txt
$B --packed 2>/dev/null | python3 tools/bench_line.py "k=31 packed" | tee $O/packed.txt
for spec in "36 400000000" "50 300000000" "62 240000000" "75 200000000" "100 150000000" "112 130000000" "125 120000000" "150 100000000" "161 93000000" "170 88000000" "200 75000000" "208 72000000" "216 69000000" "224 66000000" "250 60000000" "256 58000000" "300 50000000" "1000 15000000" "10000 1500000"; do set -- $spec
  $B --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > $O/len_sweep.txt; cat $O/len_sweep.txt
python3 tools/bench_ragged.py 100000000 31 > $O/ragged.txt 2>/dev/null; python3 tools/bench_ragged.py 100000000 21 >> $O/ragged.txt 2>/dev/null; cat $O/ragged.txt
