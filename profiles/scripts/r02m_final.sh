# final measurement pass of a build: PMC traffic (profiles/pmc_traffic.json for this libhgx.so), rocprofv3 kernel stats of the
# bench command, then the bench line (which then quotes the traffic)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python profiles/scripts/r02_pmc.py /tmp/r02m_pmc > gpurun_out/r02m_pmc.txt 2>&1
cp /tmp/r02m_pmc/kernel_stats.txt gpurun_out/r02m_pmc_driver_kernel_stats.txt 2>/dev/null
cp profiles/pmc_traffic.json gpurun_out/r02m_pmc_traffic.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02m_prof -- python "$GRAFT_REPO_ROOT/bench.py" > /tmp/r02m_prof_bench.log 2>&1 )
f=$(find /tmp/r02m_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py" > gpurun_out/r02m_kernel_stats.txt; head -40 "$f" >> gpurun_out/r02m_kernel_stats.txt; }
sleep 8  # (a process that starts while the previous GPU process is being torn down sees slow allocations: the cold leg would show them)
timeout 900 python bench.py > gpurun_out/r02m_bench.log 2> gpurun_out/r02m_bench.err
tail -c 1500 gpurun_out/r02m_pmc.txt; head -12 gpurun_out/r02m_kernel_stats.txt | cut -c1-160; tail -c 300 gpurun_out/r02m_bench.log
