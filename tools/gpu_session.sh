# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7d; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "misaligned" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
