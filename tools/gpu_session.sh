out=gpurun_out/r03d
mkdir -p $out
python3 tools/bench_windows2.py > $out/windows2_bench.txt 2>/dev/null
python3 tools/bench_ragged.py 100000000 31 > $out/ragged_bench.txt 2>/dev/null
python3 tools/bench_ragged.py 100000000 21 >> $out/ragged_bench.txt 2>/dev/null
HIST=20 python3 tools/bench_dirty.py > $out/dirty_bench.txt 2>/dev/null
python3 tools/bench_windows.py > $out/windows_bench.txt 2>/dev/null
python3 tools/bench_hist.py 100000000 12,16,20,22,23,24,26,28 > $out/hist_bench.txt 2>/dev/null
python3 tools/bench_minimizers.py > $out/minimizers_bench.txt 2>/dev/null
python3 tools/bench_fastx.py > $out/fastx_bench.txt 2>/dev/null
python3 tools/bench_fastq_pipeline.py 2>/dev/null | grep -v amdgpu.ids > $out/fastq_pipeline.txt
python3 tools/bench_elem.py 2>/dev/null | grep -v amdgpu > $out/elem_bench.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3 > $out/pytest.txt
