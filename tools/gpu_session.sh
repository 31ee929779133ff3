python -m pytest tests -x -q -m gpu -k "hist" 2>&1 | tail -3
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --sustain-steps 0 2>/dev/null | python tools/bench_line.py hist20
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hist_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --sustain-steps 0 > /dev/null 2>&1
head -4 $GRAFT_REPO_ROOT/gpurun_out/hist_trace/t_kernel_stats.csv | cut -c1-60,200-320
