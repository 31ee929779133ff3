for v in default p2t1024 p2t256 default p2t1024; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --sustain-steps 0 2>/dev/null | python tools/bench_line.py "[$v]"
done
