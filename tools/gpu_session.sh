#!/bin/bash
# scratch: the GPU session of the moment
mkdir -p gpurun_out/r6ab
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for spec in "-k 63" "-k 21" "-k 31" "-k 13" "-k 47" "-k 31 --read-len 112 --reads-per-gpu 130000000" "-k 31 --read-len 250 --reads-per-gpu 60000000" "-k 31 --read-len 170 --reads-per-gpu 88000000" "-k 31 --read-len 300 --reads-per-gpu 50000000"; do
  (cd $R/tools/_variants/r5tree && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 300 $spec 2>/dev/null | python3 $R/tools/bench_line.py "[r5  $spec]")
  (cd $R && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 300 $spec 2>/dev/null | python3 tools/bench_line.py "[r6  $spec]")
done
done 2>&1 | tee $R/gpurun_out/r6ab/ab_r5_r6_pend.txt
