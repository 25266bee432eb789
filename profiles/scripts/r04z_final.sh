#!/bin/bash
# measurement pass of a build: the whole GPU suite, PMC traffic with in-run calibration for both forms (profiles/pmc_traffic.json for
# these device sources), rocprofv3 kernel stats of the bench command in its default form (two batches in flight) and with one plan
# (--in-flight 1: the worker form), the bench line (which then quotes the traffic), the one-rank exchange self-test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04z
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 1500 python profiles/scripts/r04_pmc.py /tmp/r04z_pmc > $O/pmc.txt 2>&1
cp /tmp/r04z_pmc/kernel_stats.txt $O/pmc_driver_kernel_stats.txt 2>/dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04z_prof -- python "$GRAFT_REPO_ROOT/bench.py" > /tmp/r04z_prof_bench.log 2>&1 )
f=$(find /tmp/r04z_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py   (default: two batches in flight; every leg)" > $O/kernel_stats.txt; head -70 "$f" >> $O/kernel_stats.txt; }
L="--cfg4 0 --wide 0 --cpu-sample 0 --maf-full 0 --maf-columns 0 --columns 0 --text-path 0 --features 0 --rotating 0 --sustained-seconds 0"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04z_prof1 -- python "$GRAFT_REPO_ROOT/bench.py" --in-flight 1 $L > /tmp/r04z_prof1_bench.log 2>&1 )
f=$(find /tmp/r04z_prof1 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --in-flight 1 $L   (one plan: the worker form)" > $O/kernel_stats_one_plan.txt; head -40 "$f" >> $O/kernel_stats_one_plan.txt; }
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04z_prof2 -- python "$GRAFT_REPO_ROOT/bench.py" $L > /tmp/r04z_prof2_bench.log 2>&1 )
f=$(find /tmp/r04z_prof2 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $L   (two batches in flight: the form of value, only the liftover legs)" > $O/kernel_stats_in_flight.txt; head -40 "$f" >> $O/kernel_stats_in_flight.txt; }
sleep 5
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
timeout 300 python bench.py --exchange-selftest 1 --steps 20 --columns 0 --wide 0 --cfg4 0 --text-path 0 --cpu-sample 0 --cpu-all-cores 0 --sustained-seconds 0 --features 0 --rotating 0 > $O/bench_selftest.log 2> $O/bench_selftest.err
echo "selftest rc=$?" >> $O/bench_selftest.err
tail -3 $O/tests.log; grep -E "calibration|rotating" $O/pmc.txt | cut -c1-250; head -6 $O/kernel_stats_one_plan.txt | cut -c1-150; head -6 $O/kernel_stats_in_flight.txt | cut -c1-150; tail -1 $O/bench.err; tail -1 $O/bench_selftest.err; tail -c 300 $O/bench.log
