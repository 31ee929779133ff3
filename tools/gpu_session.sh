# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4z; mkdir -p $O
export KMX_LIB_VARIANT=bpc
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for cap in 0 3 2; do
  export KMX_TMP_BPC=$cap
  for spec in "50 300000000" "75 200000000" "100 150000000" "112 130000000" "150 100000000"; do set -- $spec
    $B --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "[cap $cap] L=$1"; done
  $B --packed 2>/dev/null | python3 tools/bench_line.py "[cap $cap] packed"
done > $O/bpc.txt 2>&1; cat $O/bpc.txt
