#!/bin/bash
# round 3, sixth GPU call: --global fix; then the measurement pass of this build: PMC traffic with in-run calibration
# (profiles/pmc_traffic.json), rocprofv3 kernel stats of the bench command, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_maxrefgap.py tests/test_gpu_columns.py -q -k "global or clones or unique" > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 1200 python profiles/scripts/r03_pmc.py /tmp/r03f_pmc > $O/pmc.txt 2>&1
cp /tmp/r03f_pmc/kernel_stats.txt $O/pmc_driver_kernel_stats.txt 2>/dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03f_prof -- python "$GRAFT_REPO_ROOT/bench.py" > /tmp/r03f_prof_bench.log 2>&1 )
f=$(find /tmp/r03f_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py" > $O/kernel_stats.txt; head -60 "$f" >> $O/kernel_stats.txt; }
sleep 8
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -5 $O/tests.log; tail -c 1500 $O/pmc.txt; head -12 $O/kernel_stats.txt | cut -c1-160; tail -c 600 $O/bench.log
