#!/bin/bash
# Round-1 closing profile of the default bench configuration (cfg2, table of the whole path):
#   gpurun_out/r01q_bench.log           python bench.py (default flags)
#   gpurun_out/r01q_bench_walk.log      the same with HGX_COMPOSED_UP=0 (level-by-level walk)
#   gpurun_out/r01q_bench_uptable.log   the same with HGX_COMPOSED_THROUGH=0 (up table + down phase)
#   gpurun_out/r01q_kernel_stats.txt    rocprofv3 --kernel-trace --stats summary of the default command
#   gpurun_out/r01q_pmc.txt             PMC passes (counters only) of the default command, all liftover kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/r01q_bench.log
HGX_COMPOSED_UP=0 timeout 300 python bench.py --columns 0 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/r01q_bench_walk.log
HGX_COMPOSED_THROUGH=0 timeout 300 python bench.py --columns 0 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/r01q_bench_uptable.log
rm -rf /tmp/r01q_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r01q_trace -- python bench.py --columns 0 --cpu-sample 0 > /tmp/r01q_trace.log 2>&1
f=$(find /tmp/r01q_trace -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --columns 0 --cpu-sample 0   (warm-up, settle and timed runs together)"; cat "$f"; } > gpurun_out/r01q_kernel_stats.txt
timeout 1200 python profiles/scripts/pmc_passes.py /tmp/r01q_pmc "" > gpurun_out/r01q_pmc.txt 2>&1
tail -5 gpurun_out/r01q_kernel_stats.txt
