# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7w; mkdir -p $O
cd $GRAFT_REPO_ROOT
KMX_LIB_VARIANT=tim python3 tools/bs_timing.py 100000000 3 2>&1 | grep -v amdgpu.ids | tee $O/timing.txt
