# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for spec in "31 300 50000000" "31 1000 15000000" "63 300 50000000" "63 1000 15000000" "41 1000 15000000" "33 10000 1500000"; do set -- $spec
  $B -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "k=$1 L=$2"; done | tee $O/long.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "headline" | tee -a $O/long.txt
