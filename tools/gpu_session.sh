mkdir -p gpurun_out/r3v
for v in default w5; do
if [ "$v" == "default" ]; then unset KMX_LIB_VARIANT; else export KMX_LIB_VARIANT=$v; fi
for spec in "36 400000000" "50 300000000" "75 200000000" "76 197000000" "80 187000000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "[$v] L=$1 k=31"
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 --config 2 -k 21 2>/dev/null | python3 tools/bench_line.py "[$v] L=$1 k=21"; done
done > gpurun_out/r3v/short.txt
unset KMX_LIB_VARIANT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3v/pytest.txt
