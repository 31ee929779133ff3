mkdir -p gpurun_out/r3h24
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3h24/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_hist.py 100000000 24 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3h24/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {} | cut -c1-260' > gpurun_out/r3h24/kernel_stats.txt
