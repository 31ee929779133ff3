# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r6s; mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for rep in 1 2; do for v in default sg2; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
for spec in "31 300 50000000" "31 1000 15000000" "31 10000 1500000"; do set -- $spec
  $B -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "[$v] k=$1 L=$2"; done; done; done | tee $O/long.txt
