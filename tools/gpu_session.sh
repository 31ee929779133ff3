# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "windows or fuzz" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
python3 tools/bench_windows.py 2>/dev/null | grep ragged
