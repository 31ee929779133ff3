#!/bin/bash
mkdir -p gpurun_out/s4
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s4/pytest.txt
tail -3 gpurun_out/s4/pytest.txt
for k in 31 21 27 29 25 13; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --sustain-steps 400 -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"
  KMX_LIB_VARIANT=r1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --sustain-steps 400 -k $k 2>/dev/null | python3 tools/bench_line.py "r1 k=$k"
done 2>&1 | tee gpurun_out/s4/k_sweep.txt
tools/pmc_pass.sh gpurun_out/s4 "SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAIT_ANY" 2>&1 | tee gpurun_out/s4/pmc.txt | head -60
