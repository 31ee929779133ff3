mkdir -p gpurun_out/r3uc
timeout 600 python tools/_uc_tmp.py > gpurun_out/r3uc/uc.txt 2>&1
