#!/bin/bash
# r06bf: rocprofv3 --kernel-trace --stats of the timed form and of the depth runs, the round's last library
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06bf
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
L="--cfg4 0 --wide 0 --cpu-sample 0 --maf-full 0 --maf-columns 0 --columns 0 --text-path 0 --features 0 --sustained-seconds 0"
( cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06bf_prof -- python $R/bench.py $L > /tmp/r06bf_prof.log 2>&1 )
f=$(find /tmp/r06bf_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $L   (two batches in flight, four rotating: the form of value)" > $O/kernel_stats_in_flight.txt; head -30 "$f" >> $O/kernel_stats_in_flight.txt; }
( cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06bf_prof2 -- python $R/profiles/scripts/column_depth_timing.py > /tmp/r06bf_prof2.log 2>&1 )
f=$(find /tmp/r06bf_prof2 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python profiles/scripts/column_depth_timing.py   (cfg2 leaf x3, --countDupes x3, root, cfg5 leaf x2)" > $O/kernel_stats_depth.txt; head -30 "$f" >> $O/kernel_stats_depth.txt; }
( cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06bf_prof3 -- python $R/bench.py --workload cfg4 --queries 1250000 $L > /tmp/r06bf_prof3.log 2>&1 )
f=$(find /tmp/r06bf_prof3 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg4 --queries 1250000 $L   (config 4's shard as the timed workload)" > $O/kernel_stats_cfg4.txt; head -20 "$f" >> $O/kernel_stats_cfg4.txt; }
ls $O
