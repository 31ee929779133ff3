# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5t; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.txt 2>&1; tail -14 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
