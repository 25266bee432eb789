#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06v
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 300 python profiles/scripts/r06v_scout_diag.py 50 > $O/diag.txt 2>&1; cat $O/diag.txt | tail -5
for m in scouts list inline; do
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06v_$m -- python $R/profiles/scripts/r06v_scout_diag.py 50 $m > /tmp/r06v_$m.log 2>&1 )
f=$(find /tmp/r06v_$m -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python profiles/scripts/r06v_scout_diag.py 50 $m" > $O/kernel_stats_$m.txt; head -12 "$f" >> $O/kernel_stats_$m.txt; }
grep "k_lift" $O/kernel_stats_$m.txt | cut -c1-60,300-
done
timeout 300 python profiles/scripts/column_depth_timing.py > $O/depth.txt 2>&1; tail -5 $O/depth.txt
