cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/rag_pmc/pmc_1 -o t -- python3 $R/tools/bench_ragged.py 100000000 31 > $R/gpurun_out/rag_pmc/pmc_1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/rag_pmc/pmc_2 -o t -- python3 $R/tools/bench_ragged.py 100000000 31 > $R/gpurun_out/rag_pmc/pmc_2.log 2>&1
cd $R && python3 tools/pmc_summary.py gpurun_out/rag_pmc > gpurun_out/rag_pmc/summary.txt
