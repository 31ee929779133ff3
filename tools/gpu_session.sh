python -m pytest tests -x -q -m gpu -k "ragged or offsets or fuzz or long" 2>&1 | tail -2
for v in default prev default prev; do
  if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
  echo "== $v"; python tools/bench_ragged.py 100000000 31 2>/dev/null | sed -n 2p; python tools/bench_ragged.py 100000000 21 2>/dev/null | sed -n 2p
done
