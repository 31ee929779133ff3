mkdir -p gpurun_out/r3w
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "eight or five or thirteen" 2>&1 | tail -3 > gpurun_out/r3w/pytest.txt
for v in default old8; do
if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
for spec in "113 132000000" "125 120000000" "128 117000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "[$v] L=$1 k=31"; done
done > gpurun_out/r3w/l8.txt
