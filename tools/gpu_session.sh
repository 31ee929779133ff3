# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in default nowib nolane seg10; do
for spec in "63 1000 15000000" "41 1000 15000000" "33 10000 1500000" "63 300 50000000" "47 500 30000000"; do set -- $spec
  python3 tools/bench_variant.py $v --no-cpu-baseline --no-traffic --sustain-steps 100 -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "[$v] k=$1 L=$2"; done; done; done > $O/seg2_variants.txt; cat $O/seg2_variants.txt
for rep in 1 2; do for v in default rrot3 rlate7 rlate3; do echo "[$v]"; KMX_DEV_LIB=$v python3 tools/bench_ragged.py 100000000 31 2>/dev/null | grep "hint 150\|100..160\|hint 160"; done; done > $O/ragged3_variants.txt; cat $O/ragged3_variants.txt
for v in default win2w; do echo "[$v]"; KMX_DEV_LIB=$v python3 tools/bench_windows.py 2>/dev/null | tail -12; done > $O/windows_2w.txt; cat $O/windows_2w.txt
