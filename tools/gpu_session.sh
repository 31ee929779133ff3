mkdir -p gpurun_out/r3fs
timeout 900 python -m pytest tests/test_gpu_fastx.py tests/test_cpp_host_layer.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3fs/pytest.txt
for v in "" nostage "" nostage; do
  echo "== variant '$v'"
  KMX_LIB_VARIANT=$v timeout 600 python tools/bench_fastq_pipeline.py 2>&1 | grep "parse" | cut -c1-60
done > gpurun_out/r3fs/stage.txt 2>&1
timeout 300 python tools/bench_fastx.py > gpurun_out/r3fs/fastx_bench.txt 2>&1
KMX_LIB_VARIANT=nostage timeout 300 python tools/bench_fastx.py > gpurun_out/r3fs/fastx_bench_nostage.txt 2>&1
