mkdir -p gpurun_out/r3lr
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_fastx.py -x -q -m gpu -k "length or fastx or reads or gate" 2>&1 | tail -3 > gpurun_out/r3lr/pytest.txt
timeout 600 python tools/bench_fastq_pipeline.py > gpurun_out/r3lr/pipeline.txt 2>&1
