# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4j; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
bash tools/variants.sh pre default > $O/variants.txt 2>&1; cat $O/variants.txt
for v in default w2; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  echo "== $v"
  for spec in "170 88000000" "200 75000000" "250 60000000" "256 58000000" "300 50000000" "1000 15000000"; do set -- $spec
    python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done
  python3 tools/bench_ragged.py 100000000 31 2>/dev/null
done > $O/waves.txt 2>&1; cat $O/waves.txt
