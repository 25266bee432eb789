cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02c_pytest.log
cat gpurun_out/r02c_pytest.log | tail -5
timeout 600 python bench.py > gpurun_out/r02c_bench.log 2> gpurun_out/r02c_bench.err
tail -c 3000 gpurun_out/r02c_bench.log; tail -5 gpurun_out/r02c_bench.err
python profiles/scripts/r02_merged_step.py 1.0 1250000 cfg4 cfg4 > gpurun_out/r02c_cfg4.log 2>&1; grep -v amdgpu.ids gpurun_out/r02c_cfg4.log
