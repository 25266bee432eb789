cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02s_gpu_tests.txt
cat gpurun_out/r02s_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash profiles/scripts/r02m_final.sh > gpurun_out/r02s_final.txt 2>&1
tail -5 gpurun_out/r02s_final.txt | cut -c1-300
