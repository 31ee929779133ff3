mkdir -p gpurun_out/r3fin
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3fin/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3fin/smoke.txt 2>&1
timeout 600 python tools/bench_fastq_pipeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r3fin/fastq_pipeline.txt
