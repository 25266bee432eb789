#!/bin/bash
# r06 call 2: config 3 (hal2maf over the whole reference) as ONE bench leg with the export's timing lines, then the same under rocprofv3.
#   gpurun --timeout 900 -- 'bash profiles/scripts/r06b_cfg3_leg.sh'
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06b
mkdir -p $O
export TMPDIR=/tmp
HGX_MAF_TIMING=1 timeout 400 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/1_cfg3_leg.json 2> $O/1_cfg3_timing.txt; echo "cfg3 leg rc=$?" | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/prof_cfg3" -- python "$OLDPWD/bench.py" --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > "$OLDPWD/$O/2_rocprof.txt" 2>&1); echo "rocprof rc=$?" | tee -a $O/summary.txt
find $O/prof_cfg3 -name "*kernel_stats*" -exec cp {} $O/2_kernel_stats.csv \; 2>/dev/null
rm -rf $O/prof_cfg3
tail -c 3000 $O/1_cfg3_leg.json
tail -n 30 $O/1_cfg3_timing.txt
