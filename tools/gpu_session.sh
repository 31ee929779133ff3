# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
KMX_FUZZ_N=12000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_fuzz12000.txt 2>&1; tail -2 $O/pytest_fuzz12000.txt
rm -rf gpurun_out/r05; bash tools/profile_round.sh gpurun_out/r05 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
