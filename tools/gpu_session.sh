mkdir -p gpurun_out/r3el
timeout 600 python tools/bench_elem.py > gpurun_out/r3el/elem2.txt 2>&1
timeout 1500 python -m pytest tests/ -x -q -m gpu -k "word or revcomp or canonical_words or hash_words or kats or cpp" 2>&1 | tail -3 > gpurun_out/r3el/pytest.txt
