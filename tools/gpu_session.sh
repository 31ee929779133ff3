mkdir -p gpurun_out/r3fx
timeout 900 python -m pytest tests/test_gpu_fastx.py tests/test_cpp_host_layer.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3fx/pytest.txt
timeout 600 python tools/bench_fastq_pipeline.py > gpurun_out/r3fx/pipeline.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3fx/prof2 -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_fastq_pipeline.py > $GRAFT_REPO_ROOT/gpurun_out/r3fx/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3fx/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep fastx {} | cut -c1-200' > gpurun_out/r3fx/kernel_stats.txt
