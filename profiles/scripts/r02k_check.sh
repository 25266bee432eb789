#!/bin/bash
# parity subset + step time of the cfg2 / cfg4 batches
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_composed.py tests/test_gpu_altpaths.py tests/test_gpu_configs.py tests/test_gpu_limits.py tests/test_gpu_liftover.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02k_tests.txt
cat gpurun_out/r02k_tests.txt
( timeout 300 python profiles/scripts/r02_merged_step.py 1.0 1000000 cfg2 default
  HGX_FINISH_WAVE=0 timeout 300 python profiles/scripts/r02_merged_step.py 1.0 1000000 cfg2 lds
  timeout 300 python profiles/scripts/r02_merged_step.py 1.0 1000000 cfg4 default4
  HGX_LIB_PATH=/root/repo/hal_amd/libhgx_prof.so python profiles/scripts/r02_merged_step.py 1.0 1000000 cfg2 prof 2>&1 | grep "lift profile\|prof " | tail -3 ) 2>&1 | grep -v "^$\|amdgpu.ids" > gpurun_out/r02k_step.txt
cat gpurun_out/r02k_step.txt
