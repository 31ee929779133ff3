# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in default seg3a seg3b; do
for spec in "33 10000 1500000" "36 1000 15000000" "37 1000 15000000" "41 1000 15000000" "47 500 30000000" "41 300 50000000"; do set -- $spec
  python3 tools/bench_variant.py $v --no-cpu-baseline --no-traffic --sustain-steps 100 -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "[$v] k=$1 L=$2"; done; done; done > $O/seg2_3waves.txt; cat $O/seg2_3waves.txt
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_fuzz.py -x -q -m gpu -k "two_word or segments or fuzz or reduce2" --kmx-lib tools/_variants/seg3b/libkmx.so > $O/pytest_seg3b.txt 2>&1; tail -2 $O/pytest_seg3b.txt
