mkdir -p gpurun_out/r3aux
tools/variants.sh default aux0 aux1 aux3 aux16 aux17 aux18 aux19 > gpurun_out/r3aux/variants.txt 2>&1
