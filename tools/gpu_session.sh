#!/bin/bash
# scratch: the GPU session of the moment
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python3 tools/bench_fastq_pipeline.py > gpurun_out/r05_fastq_pipeline.txt 2>&1; cat gpurun_out/r05_fastq_pipeline.txt
timeout 300 python3 tools/bench_fastx.py > gpurun_out/r05_fastx_bench.txt 2>&1; cat gpurun_out/r05_fastx_bench.txt
bash tools/pmc_fastq.sh gpurun_out/pmc_fastq3 "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU2" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" > /dev/null 2>&1
timeout 120 python3 tools/bench_fastq_parse.py 256 12 > gpurun_out/fq_parse.txt 2>&1; tail -1 gpurun_out/fq_parse.txt
