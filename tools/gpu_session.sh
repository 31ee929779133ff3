# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 2400 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu --kmx-lib tools/_variants/r3x/libkmx.so > $O/pytest_r3x.txt 2>&1; tail -2 $O/pytest_r3x.txt
bash tools/variants.sh default nowib > $O/headline_wib.txt 2>&1; cat $O/headline_wib.txt
for spec in "1000 15000000" "300 50000000" "250 60000000" "256 58000000" "200 75000000" "100 150000000" "50 300000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > $O/len_some.txt; cat $O/len_some.txt
for rep in 1 2; do for v in default r3x; do echo "[$v]"; KMX_DEV_LIB=$v python3 tools/bench_ragged.py 100000000 31 2>/dev/null | grep -v "^=="; done; done > $O/ragged_r3x.txt; cat $O/ragged_r3x.txt
for v in default r3x; do echo "[$v]"; KMX_DEV_LIB=$v python3 tools/bench_long_ragged.py 6e9 31 2>/dev/null | grep segments; done > $O/long_ragged_r3x.txt; cat $O/long_ragged_r3x.txt
