#!/bin/bash
# scratch: the GPU session of the moment
for v in "" fx_t512 fx_r16 fx_r4 "" fx_t512 fx_r16 fx_r4; do
  echo "== variant '$v'"
  KMX_DEV_LIB=$v timeout 120 python3 tools/bench_fastq_parse.py 256 12 2>&1 | tail -1
done
timeout 600 python3 -m pytest tests/test_gpu_fastx.py -x -q --kmx-lib tools/_variants/fx_t512/libkmx.so 2>&1 | tail -3
