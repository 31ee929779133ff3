./tools/ubench9 > gpurun_out/ubench9.txt; ./tools/ubench10 > gpurun_out/ubench10.txt
