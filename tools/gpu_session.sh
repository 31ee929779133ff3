# scratch: the command list of the last gpurun call (round 3: the profile set of the final build + the full GPU suite)
mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r03/pytest_gpu.txt
bash tools/profile_round.sh gpurun_out/r03 > gpurun_out/r03/profile_round.log 2>&1
