# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for spec in "63 200 75000000" "63 250 60000000" "63 300 50000000" "63 1000 15000000" "41 1000 15000000" "47 500 30000000" "33 10000 1500000" "63 10000 1500000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "k=$1 L=$2"; done > $O/k2_long.txt; cat $O/k2_long.txt
(echo "## uniform 150 bp (instrumented build)"; KMX_DEV_LIB=bst python3 tools/bs_timing.py 100000000 3
 echo "## 2 % trimmed, bound 150, ragged kernel at 2 waves"; KMX_DEV_LIB=bst python3 tools/bs_timing.py 100000000 3 trim
 echo "## 2 % trimmed, bound 150, ragged kernel at 3 waves"; KMX_DEV_LIB=bst3 python3 tools/bs_timing.py 100000000 3 trim) 2>/dev/null > $O/phase_timing.txt; cat $O/phase_timing.txt
KMX_PMC_BENCH_ARGS="--config 4" bash tools/pmc_pass.sh gpurun_out/r5e/pmc_hist "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/pmc_summary.py gpurun_out/r5e/pmc_hist 2>/dev/null | grep -v "^==.*kernel_stats" | grep -B1 -A12 "SinkHistPart\|hist_part_reduce\|calib" | head -150 > $O/pmc_hist_summary.txt; cat $O/pmc_hist_summary.txt
