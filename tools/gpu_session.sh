mkdir -p gpurun_out/r3rg
for v in "" rag3t rag4 "" rag4; do
  echo "== variant '$v'"
  KMX_BS_PRINT_BPC=1 KMX_LIB_VARIANT=$v timeout 600 python tools/bench_ragged.py 100000000 31 2>&1 | grep -v amdgpu
done > gpurun_out/r3rg/ragged.txt 2>&1
KMX_LIB_VARIANT=rag4 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py -x -q -m gpu -k "ragged or offsets or rolled or fuzz" 2>&1 | tail -3 > gpurun_out/r3rg/pytest.txt
