# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5u; mkdir -p $O
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r05; bash tools/profile_round.sh gpurun_out/r05 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
