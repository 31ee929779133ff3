R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 0"
rm -rf $out/trace_hist20
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_hist20 -o t -- $B --config 4 --steps 5 --warmup 2 > $out/bench_hist20_under_trace.json 2> /dev/null
cd $R
python3 bench.py --config 4 --no-traffic --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 20 > $out/bench_hist20.json 2> /dev/null
python3 bench.py --config 4 --dist-single --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 0 > $out/bench_hist20_rccl1.json 2> /dev/null
KMX_BENCH_TEST_SHARED_GPU=1 python3 bench.py --gpus 2 --config 4 --reads-per-gpu 20000000 --steps 3 --warmup 1 --sustain-steps 0 --no-traffic --cpu-baseline-seconds 8 > $out/bench_2ranks_shared_gpu_hist20.json 2> /dev/null
for spec in "36 400000000" "50 300000000" "62 240000000" "75 200000000" "100 150000000" "125 120000000" "150 100000000" "161 93000000" "170 88000000" "200 75000000" "208 72000000" "250 60000000" "256 58000000" "300 50000000" "1000 15000000" "10000 1500000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > $out/len_sweep.txt
python3 tools/bench_hist.py 100000000 12,16,20,22,23,24,26,28 > $out/hist_bench.txt 2>/dev/null
