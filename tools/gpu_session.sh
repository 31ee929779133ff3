# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r8a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "k=31"
python3 bench.py --config 2 -k 63 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "k=63"
python3 bench.py --config 2 -k 21 --no-cpu-baseline --no-traffic 2>/dev/null | python3 tools/bench_line.py "k=21"
python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len 1000 --reads-per-gpu 15000000 2>/dev/null | python3 tools/bench_line.py "L=1000"
python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len 100 --reads-per-gpu 150000000 2>/dev/null | python3 tools/bench_line.py "L=100"
python3 tools/bench_ragged.py 100000000 31 2>/dev/null | head -7
