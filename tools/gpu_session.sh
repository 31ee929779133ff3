# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7v; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "behind_offsets or add_up" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
