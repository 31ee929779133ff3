# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7j; mkdir -p $O
cd $GRAFT_REPO_ROOT
for spec in "5000000 300" "1500000 1000" "150000 10000"; do set -- $spec; echo "[$1 reads of $2 bases: planned as segments on the device]"; python3 tools/bench_windows.py $1 $2 2>/dev/null | grep "^k="; done | tee $O/windows_long.txt
python3 tools/bench_windows2.py > $O/windows2_bench.txt 2>/dev/null; tail -4 $O/windows2_bench.txt
KMX_FUZZ_N=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
