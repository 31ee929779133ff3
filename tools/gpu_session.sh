# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r8f; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
KMX_FUZZ_N=6000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "driver command"; done
