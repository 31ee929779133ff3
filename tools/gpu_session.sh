# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5x; mkdir -p $O
cd $GRAFT_REPO_ROOT
KMX_FUZZ_N=30000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_fuzz30000.txt 2>&1; tail -3 $O/pytest_fuzz30000.txt
