# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r6n; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "hist" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for v in default hp0; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  echo "[$v]"; python3 tools/bench_hist.py 100000000 23,24,28 2>/dev/null; done; done | tee $O/hist.txt
