mkdir -p gpurun_out/r3fz
timeout 900 python tools/stress_contexts.py > gpurun_out/r3fz/stress.txt 2>&1; echo "rc=$?" >> gpurun_out/r3fz/stress.txt
