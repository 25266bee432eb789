cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r02i_bench.log 2> gpurun_out/r02i_bench.err
tail -c 2500 gpurun_out/r02i_bench.log; tail -3 gpurun_out/r02i_bench.err
timeout 1200 python profiles/scripts/r02_pmc.py /tmp/r02i_pmc > gpurun_out/r02i_pmc.txt 2>&1
cp /tmp/r02i_pmc/kernel_stats.txt gpurun_out/r02i_kernel_stats.txt 2>/dev/null
cp profiles/pmc_traffic.json gpurun_out/r02i_pmc_traffic.json
cat gpurun_out/r02i_pmc.txt | tail -40
