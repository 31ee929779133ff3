python -m pytest tests -x -q -m gpu -k "hist" 2>&1 | tail -2
for v in default hrot0 default hrot0; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --sustain-steps 0 2>/dev/null | python tools/bench_line.py "[$v]"
done
