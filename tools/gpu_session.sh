mkdir -p gpurun_out/r3q
for v in default w13_3; do
if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
for spec in "161 93000000" "170 88000000" "185 81000000" "200 75000000" "208 72000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "[$v] L=$1"; done
done > gpurun_out/r3q/len13.txt
unset KMX_LIB_VARIANT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lengths or uniform" 2>&1 | tail -3 > gpurun_out/r3q/pytest.txt
