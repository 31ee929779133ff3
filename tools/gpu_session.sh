mkdir -p gpurun_out/r3z
tools/variants.sh default xorc swz2 xs > gpurun_out/r3z/variants.txt 2>&1
KMX_LIB_VARIANT=xs python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3z/pytest_xs.txt
