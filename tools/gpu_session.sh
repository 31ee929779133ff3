bash tools/profile_round.sh gpurun_out/r03b > gpurun_out/r03b.log 2>&1
