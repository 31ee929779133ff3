# scratch: the command list of the last gpurun call (round 3, run 11: device-side uniform/ragged gate, ragged buffer loads)
mkdir -p gpurun_out/r3k
python -m pytest tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3k/pytest_r3.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3k/pytest_all.txt
python3 tools/bench_ragged.py 100000000 31 > gpurun_out/r3k/ragged31.txt 2>&1
python3 tools/bench_ragged.py 100000000 21 > gpurun_out/r3k/ragged21.txt 2>&1
