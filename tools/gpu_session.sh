#!/bin/bash
for v in default h4 h2 h16; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  echo "== $v"; python3 tools/bench_hist.py 100000000 16,20 2>/dev/null
done
