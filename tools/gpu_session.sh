# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
KMX_FUZZ_N=12000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_fuzz12000.txt 2>&1; tail -2 $O/pytest_fuzz12000.txt
python3 tools/bench_ragged.py 100000000 31 2>/dev/null > $O/ragged_bench.txt; cat $O/ragged_bench.txt
python3 tools/bench_ragged2.py 100000000 2>/dev/null > $O/ragged2_bench.txt; cat $O/ragged2_bench.txt
python3 tools/bench_fastq_pipeline.py 2>/dev/null | grep -v amdgpu.ids > $O/fastq_pipeline.txt; cat $O/fastq_pipeline.txt
python3 bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "[final]"
