#!/bin/bash
# dev tool (run on the GPU box through gpurun): the profile set committed under profiles/ for one round.
#   tools/profile_round.sh OUTDIR      -> OUTDIR/{trace*,pmc_*}/..., OUTDIR/bench_*.json, OUTDIR/*.txt
# rocprofv3 runs from /tmp; --pmc passes are separate runs without any tracing flags.
R=$GRAFT_REPO_ROOT
out=$R/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 0"
# kernel traces: the headline config and the other BASELINE configs
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/bench_under_trace.json 2> $out/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_k21 -o t -- $B --config 2 -k 21 > $out/bench_k21_under_trace.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_k63 -o t -- $B --config 2 -k 63 > $out/bench_k63_under_trace.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_hash -o t -- $B --config 3 > $out/bench_hash_under_trace.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_hist20 -o t -- $B --config 4 --steps 5 --warmup 2 > $out/bench_hist20_under_trace.json 2> /dev/null
# counters (the bench's own two --pmc children cover FETCH_SIZE / WRITE_SIZE with the calibration kernel; these are the SQ sets)
for c in "SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$tag -o t -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-traffic --sustain-steps 0 > $out/pmc_$tag.json 2> $out/pmc_$tag.err
done
cd $R
python3 tools/pmc_summary.py $out > $out/pmc_summary.txt 2>&1
# un-profiled lines: the driver's command (with in-run traffic + CPU baseline), the other configs, the sustained run
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 3000 > $out/bench_sustained3000.json 2> /dev/null
python3 bench.py --config 2 -k 21 --no-cpu-baseline > $out/bench_k21.json 2> /dev/null
python3 bench.py --config 2 -k 63 --no-cpu-baseline > $out/bench_k63.json 2> /dev/null
python3 bench.py --config 3 --no-cpu-baseline > $out/bench_hash.json 2> /dev/null
python3 bench.py --config 4 --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 20 > $out/bench_hist20.json 2> /dev/null
python3 bench.py --config 4 --dist-single --no-cpu-baseline --steps 5 --warmup 2 --sustain-steps 0 > $out/bench_hist20_rccl1.json 2> /dev/null
# the N > 1 control flow on this one-GPU box (both ranks on cuda:0, gloo; stated in the line): what an 8-GPU line will carry
KMX_BENCH_TEST_SHARED_GPU=1 python3 bench.py --gpus 2 --reads-per-gpu 20000000 --steps 5 --warmup 2 --sustain-steps 0 --no-traffic --cpu-baseline-seconds 8 > $out/bench_2ranks_shared_gpu.json 2> /dev/null
KMX_BENCH_TEST_SHARED_GPU=1 python3 bench.py --gpus 2 --config 4 --reads-per-gpu 20000000 --steps 3 --warmup 1 --sustain-steps 0 --no-traffic --cpu-baseline-seconds 8 > $out/bench_2ranks_shared_gpu_hist20.json 2> /dev/null
python3 bench.py --packed --no-cpu-baseline --no-traffic > $out/bench_packed.json 2> /dev/null
for k in 9 11 12 13 17 21 25 27 29 31 33 41 47 51 55 63; do python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 200 -k $k 2>/dev/null | python3 tools/bench_line.py "k=$k"; done > $out/k_sweep.txt
for spec in "36 400000000" "50 300000000" "62 240000000" "75 200000000" "100 150000000" "112 130000000" "125 120000000" "150 100000000" "161 93000000" "170 88000000" "200 75000000" "208 72000000" "216 69000000" "224 66000000" "250 60000000" "256 58000000" "300 50000000" "1000 15000000" "10000 1500000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 --read-len $1 --reads-per-gpu $2 2>/dev/null | python3 tools/bench_line.py "L=$1"; done > $out/len_sweep.txt
# two-word k on the long frames (round 4), and the settle-steps cross-check of the headline (30 = the default, 0 beside it)
for spec in "63 200 75000000" "63 250 60000000" "47 208 72000000" "33 250 60000000" "63 300 50000000" "63 1000 15000000" "41 1000 15000000" "33 10000 1500000"; do set -- $spec
  python3 bench.py --no-cpu-baseline --no-traffic --sustain-steps 100 -k $1 --read-len $2 --reads-per-gpu $3 2>/dev/null | python3 tools/bench_line.py "k=$1 L=$2"; done > $out/k2_long.txt
for st in 30 0 30 0; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 1000 --settle-steps $st 2>/dev/null | python3 tools/bench_line.py "[settle $st]"; done > $out/settle.txt
python3 tools/bench_windows2.py > $out/windows2_bench.txt 2>/dev/null
python3 tools/bench_ragged.py 100000000 31 > $out/ragged_bench.txt 2>/dev/null
python3 tools/bench_ragged.py 100000000 21 >> $out/ragged_bench.txt 2>/dev/null
python3 tools/bench_ragged2.py 100000000 > $out/ragged2_bench.txt 2>/dev/null
HIST=20 python3 tools/bench_dirty.py > $out/dirty_bench.txt 2>/dev/null
python3 tools/bench_windows.py > $out/windows_bench.txt 2>/dev/null
for L in 100 140 150 158 166 200 250 256; do echo "[1e7 reads of $L bases]"; python3 tools/bench_windows.py 10000000 $L 2>/dev/null | grep "^k="; done > $out/windows_len.txt
for spec in "5000000 300" "1500000 1000" "150000 10000"; do set -- $spec; echo "[$1 reads of $2 bases: planned as segments on the device]"; python3 tools/bench_windows.py $1 $2 2>/dev/null | grep "^k="; done >> $out/windows_len.txt
python3 tools/bench_hist.py 100000000 12,16,20,22,23,24,26,28 > $out/hist_bench.txt 2>/dev/null
python3 tools/bench_minimizers.py > $out/minimizers_bench.txt 2>/dev/null
python3 tools/bench_fastx.py > $out/fastx_bench.txt 2>/dev/null
python3 tools/bench_fastq_pipeline.py 2>/dev/null | grep -v amdgpu.ids > $out/fastq_pipeline.txt
python3 tools/step_times.py > $out/step_times_cold.txt 2>/dev/null
python3 tools/bench_small_batches.py 2>/dev/null | grep -v amdgpu.ids > $out/small_batches.txt
ls $out
