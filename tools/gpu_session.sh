# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7u; mkdir -p $O
cd $GRAFT_REPO_ROOT
bash tools/variants.sh default st50 st90 | tee $O/static.txt
