# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4n; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for v in default w3; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  echo "== $v"
  $B 2>/dev/null | python3 tools/bench_line.py "k=31"
  $B --hash 2>/dev/null | python3 tools/bench_line.py "k=31 hash"
  for k in 49 55 63; do $B -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done
  for spec in "50 300000000" "75 200000000" "100 150000000" "216 69000000" "224 66000000" "250 60000000" "256 58000000" "300 50000000" "1000 15000000"; do set -- $spec
    $B --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done
  python3 tools/bench_ragged.py 100000000 31 2>/dev/null
done > $O/waves.txt 2>&1; cat $O/waves.txt
