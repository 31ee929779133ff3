# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4l; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100"
for k in 41 47 48 49 55 63; do $B -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done > $O/k2.txt 2>&1; cat $O/k2.txt
