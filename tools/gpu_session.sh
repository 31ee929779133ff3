# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
cd $GRAFT_REPO_ROOT
# 1. the build that returned the wrong sum in round 4 (f642082 with the three-wave ragged frame and the rotation back in)
(cd tools/_variants/old_tree && for m in "L=1000 n=150000" "L=300 n=400000" "L=1000 n=150000 mode=ragged2L" "L=150 n=2000000 mode=trim"; do timeout 900 python3 dev_bisect_sum.py $m; done) > $O/old_tree_bisect.txt 2>&1
tail -40 $O/old_tree_bisect.txt
# 2. today's source with the ragged 10-word frame at three waves
for m in "L=1000 n=150000 mode=ragged2L" "L=300 n=400000 mode=ragged2L" "L=150 n=2000000 mode=trim" "L=100 n=3000000 mode=trim"; do KMX_DEV_LIB=r3w timeout 900 python3 tools/dev_bisect_sum.py $m; done > $O/r3w_bisect.txt 2>&1
tail -30 $O/r3w_bisect.txt
# 3. the new full-input oracle tests, then the whole suite on the macro-free build
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k full_input --durations=5 > $O/pytest_full_input.txt 2>&1; tail -12 $O/pytest_full_input.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 tools/bench_line.py "[default]" < $O/bench_default.json
