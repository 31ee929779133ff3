#!/bin/bash
# dev tool (GPU box): bench the development variants of libkmx built beforehand on the CPU box with
#   python -m kmers_amd.build NAME [--only a.hip,b.hip] -DSWITCH=1 ...      -> kmers_amd/libkmx_NAME.so
# (the default libkmx.so is never touched; KMX_LIB_VARIANT selects the library at load time).
#   tools/variants.sh [bench.py args --] NAME... ; NAME "default" = libkmx.so; each variant is run twice, interleaved
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
if [ "$1" == "--" ]; then shift; else set -- "${args[@]}"; args=(); fi
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
    timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --sustain-steps 400 "${args[@]}" 2>/dev/null | python3 tools/bench_line.py "[$v]"
  done
done
unset KMX_LIB_VARIANT
