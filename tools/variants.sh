#!/bin/bash
# dev tool (GPU box): bench the development variants of libkmx built beforehand on the CPU box with
#   python tools/dev_variant.py NAME --only a.hip[,b.hip] [--sub 'REGEX=>TEXT'] [--patch FILE] [-DX=1]   -> tools/_variants/NAME/libkmx.so
# (the product's libkmx.so is never touched and never looks for a variant: tools/bench_variant.py points the loader at one).
#   tools/variants.sh [bench.py args --] NAME... ; NAME "default" = kmers_amd/libkmx.so; each variant is run twice, interleaved
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
if [ "$1" == "--" ]; then shift; else set -- "${args[@]}"; args=(); fi
for rep in 1 2; do
  for v in "$@"; do
    timeout 600 python3 tools/bench_variant.py $v --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --sustain-steps 400 "${args[@]}" 2>/dev/null | python3 tools/bench_line.py "[$v]"
  done
done
