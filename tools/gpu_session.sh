mkdir -p gpurun_out/r3nt
for v in "" mnt "" mnt; do
  echo "== variant '$v'"
  KMX_LIB_VARIANT=$v timeout 600 python tools/bench_minimizers.py 2>&1 | grep -v amdgpu | cut -c1-120
done > gpurun_out/r3nt/mnt.txt 2>&1
