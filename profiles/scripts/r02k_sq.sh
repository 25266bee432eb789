cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python profiles/scripts/r02k_sq.py /tmp/r02k_sq > gpurun_out/r02k_sq.txt 2>&1
cat gpurun_out/r02k_sq.txt | tail -30
