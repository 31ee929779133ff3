# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh gpurun_out/r04d > gpurun_out/r04d.log 2>&1
tail -3 gpurun_out/r04d.log
