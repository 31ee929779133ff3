python tools/bench_fastq_pipeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fastq_pipeline.txt
