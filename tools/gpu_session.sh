#!/bin/bash
# scratch: the GPU session of the moment
for v in "" fx_nopartial fx_nofull "" fx_nopartial fx_nofull; do
  echo "== variant '$v'"
  KMX_DEV_LIB=$v timeout 120 python3 tools/bench_fastq_parse.py 256 12 2>&1 | tail -1
done
