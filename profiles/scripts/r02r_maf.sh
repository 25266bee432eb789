mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_columns.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3
python profiles/scripts/r02n_maf_timing.py 2>&1 | grep -v amdgpu.ids | tail -4
