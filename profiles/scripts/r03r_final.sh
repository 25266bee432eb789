#!/bin/bash
# measurement pass of a build: PMC traffic with in-run calibration (profiles/pmc_traffic.json for these device sources), rocprofv3
# kernel stats of the bench command, the bench line (which then quotes the traffic), the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03r
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 1200 python profiles/scripts/r03_pmc.py /tmp/r03r_pmc > $O/pmc.txt 2>&1
cp /tmp/r03r_pmc/kernel_stats.txt $O/pmc_driver_kernel_stats.txt 2>/dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03r_prof -- python "$GRAFT_REPO_ROOT/bench.py" > /tmp/r03r_prof_bench.log 2>&1 )
f=$(find /tmp/r03r_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py" > $O/kernel_stats.txt; head -60 "$f" >> $O/kernel_stats.txt; }
sleep 8
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
timeout 300 python bench.py --exchange-selftest 1 --steps 20 --columns 0 --wide 0 --cfg4 0 --text-path 0 --cpu-sample 0 --cpu-all-cores 0 --sustained-seconds 0 > $O/bench_selftest.log 2> $O/bench_selftest.err
echo "selftest rc=$?" >> $O/bench_selftest.err
tail -3 $O/tests.log; grep calibration $O/pmc.txt | cut -c1-300; head -8 $O/kernel_stats.txt | cut -c1-150; tail -1 $O/bench.err; tail -1 $O/bench_selftest.err; tail -c 400 $O/bench.log
