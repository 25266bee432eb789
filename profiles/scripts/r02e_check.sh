cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_textpath.py tests/test_gpu_liftover.py tests/test_gpu_multiseq.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -12
timeout 900 python bench.py > gpurun_out/r02e_bench.log 2> gpurun_out/r02e_bench.err
tail -c 6000 gpurun_out/r02e_bench.log; tail -3 gpurun_out/r02e_bench.err
