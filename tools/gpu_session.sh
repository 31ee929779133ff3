# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r8g; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in default ls0 default ls0; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  for spec in "62 240000000" "75 200000000" "100 150000000" "112 130000000" "125 120000000" "161 93000000" "200 75000000" "224 66000000" "250 60000000" "256 58000000"; do set -- $spec
    python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "[$v] L=$1"; done
done | cut -c1-110 | tee $O/ls0.txt
