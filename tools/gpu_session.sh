# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4i; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_line.py "[k31]" < $O/bench_default.json
