# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
bash tools/profile_round.sh gpurun_out/r04
