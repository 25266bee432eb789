cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r02l_tests.txt
cat gpurun_out/r02l_tests.txt
timeout 900 python bench.py > gpurun_out/r02l_bench.log 2> gpurun_out/r02l_bench.err
tail -c 6000 gpurun_out/r02l_bench.log; tail -3 gpurun_out/r02l_bench.err
