# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5v; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "hist" --kmx-lib tools/_variants/hbfe/libkmx.so > $O/pytest_hbfe.txt 2>&1; tail -2 $O/pytest_hbfe.txt
for rep in 1 2 3; do for v in default hbfe; do echo "[$v]"; KMX_DEV_LIB=$v python3 tools/bench_hist.py 100000000 16,20,22,24 2>/dev/null; done; done > $O/hist_bfe.txt; cat $O/hist_bfe.txt
