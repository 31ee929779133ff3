# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r6l; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "windows or fuzz" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for L in 150 166 200 250 256; do echo "[n=1e7 L=$L]"; python3 tools/bench_windows.py 10000000 $L 2>/dev/null | grep "^k="; done | tee $O/win12.txt
