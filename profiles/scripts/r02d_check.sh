cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_composed.py tests/test_gpu_liftover.py tests/test_gpu_blockmap.py -x -q -m gpu 2>&1 | tail -5
HGX_COMPOSED_UP=1 SOAK_SEED=7 timeout 400 python profiles/scripts/soak_parity.py 150 2>&1 | tail -3 | tee gpurun_out/r02d_soak_merged.log
python profiles/scripts/r02_plan_modes.py 1.0 1000000 2>&1 | grep -v amdgpu | tee gpurun_out/r02d_modes.log
python profiles/scripts/r02_merged_step.py 1.0 1250000 cfg4 cfg4 2>&1 | grep -v amdgpu | tee gpurun_out/r02d_cfg4.log
HGX_FINISH_WAVE=0 python profiles/scripts/r02_merged_step.py 1.0 1250000 cfg4 cfg4-nowave 2>&1 | grep -v amdgpu | tee -a gpurun_out/r02d_cfg4.log
