mkdir -p gpurun_out/r3a1
tools/variants.sh default a1 > gpurun_out/r3a1/variants.txt 2>&1
