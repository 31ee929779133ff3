#!/bin/bash
# dev tool (run on the GPU box through gpurun): the profile set committed under profiles/ for one round.
#   tools/profile_round.sh OUTDIR      -> OUTDIR/{trace,pmc_*}/..., OUTDIR/bench_under_trace.json, OUTDIR/ubench*.txt
# rocprofv3 runs from /tmp; --pmc passes are separate runs without any tracing flags.
R=$GRAFT_REPO_ROOT
out=$R/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/bench_under_trace.json 2> $out/trace.err
for c in FETCH_SIZE WRITE_SIZE "SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$tag -o t -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $out/pmc_$tag.json 2> $out/pmc_$tag.err
done
cd $R
for u in ubench3 ubench4 ubench7; do [ -x tools/$u ] && ./tools/$u > $out/$u.txt 2>&1; done
python3 tools/pmc_summary.py $out > $out/pmc_summary.txt 2>&1
python3 bench.py --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err
ls $out
