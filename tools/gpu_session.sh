# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7p; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_$i.txt 2>&1; tail -1 $O/pytest_$i.txt; done
