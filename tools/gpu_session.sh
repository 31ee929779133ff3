mkdir -p gpurun_out/r3nt
for v in "" snt "" snt; do
  echo "== variant '$v'"
  KMX_LIB_VARIANT=$v timeout 600 python tools/bench_hist.py 100000000 12,20 2>&1 | grep -v amdgpu | cut -c1-100
  KMX_LIB_VARIANT=$v timeout 600 python tools/bench_windows.py 2>&1 | grep -v amdgpu | head -2 | cut -c1-100
done > gpurun_out/r3nt/snt.txt 2>&1
