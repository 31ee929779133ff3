# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r6v; mkdir -p $O
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/stream_lds_dma tools/stream_lds_dma.hip 2>/dev/null
timeout 300 /tmp/stream_lds_dma | tee $O/lds_dma.txt
