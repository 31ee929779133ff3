#!/bin/bash
# scratch: the GPU session of the moment
mkdir -p gpurun_out/r6ab
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  (cd $R/tools/_variants/r5tree && python3 bench.py --config 4 --no-cpu-baseline --no-traffic --steps 5 --warmup 2 --sustain-steps 20 2>/dev/null | python3 $R/tools/bench_line.py "[r5 hist20]")
  for v in default hsync hnomark hsyncnomark; do
  python3 tools/bench_variant.py $v --config 4 --no-cpu-baseline --no-traffic --steps 5 --warmup 2 --sustain-steps 20 2>/dev/null | python3 tools/bench_line.py "[$v hist20]"
  done
done 2>&1 | tee $R/gpurun_out/r6ab/ab_hist2.txt
