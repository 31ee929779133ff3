# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "segments" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for spec in "31 300 50000000" "31 400 37000000" "31 1000 15000000" "31 10000 1500000" "63 1000 15000000"; do set -- $spec
  $B -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "k=$1 L=$2"; done | tee $O/long.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-traffic --sustain-steps 0 --steps 10 --read-len 10000 --reads-per-gpu 1500000 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; head -6 $O/trace/t_kernel_stats.csv | cut -c1-160
