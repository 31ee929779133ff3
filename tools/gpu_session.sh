mkdir -p gpurun_out/r3nt
for v in "" fxl "" fxl; do
  echo "== variant '$v'"
  KMX_LIB_VARIANT=$v timeout 600 python tools/bench_fastq_pipeline.py 2>&1 | grep "parse" | cut -c1-60
  KMX_LIB_VARIANT=$v timeout 600 python tools/bench_fastx.py 2>&1 | grep -v amdgpu | cut -c1-150
done > gpurun_out/r3nt/fxl.txt 2>&1
