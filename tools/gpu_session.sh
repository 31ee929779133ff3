mkdir -p gpurun_out/r3fin
timeout 600 python tools/bench_fastq_pipeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r3fin/fastq_pipeline2.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3fin/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_fastq_pipeline.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3fin/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep fastx {} | cut -c1-200' > gpurun_out/r3fin/kernel_stats.txt
