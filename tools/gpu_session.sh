#!/bin/bash
mkdir -p gpurun_out/s17
tools/variants.sh default L2 L3 L4 P3 L3P3 > gpurun_out/s17/variants.txt 2>&1
cat gpurun_out/s17/variants.txt
