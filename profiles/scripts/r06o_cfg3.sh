#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06o
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_gpu_columns.py tests/test_gpu_unique.py tests/test_gpu_zz_round5.py tests/test_gpu_maxrefgap.py -p no:cacheprovider --timeout 600 > $O/1_tests.txt 2>&1; echo "column tests rc=$?" | tee $O/summary.txt
tail -n 3 $O/1_tests.txt
export R06F_EXTRA='[{},{},{"HGX_MAF_RENDERS_IN_FLIGHT":"6"}]'
timeout 600 python profiles/scripts/r06f_cfg3_sweep.py > $O/2_sweep.jsonl 2> $O/2_sweep.err; echo "sweep rc=$?" | tee -a $O/summary.txt
cat $O/2_sweep.jsonl | cut -c1-600
grep -E "CPU seconds" $O/2_sweep.err | cut -c1-460
(cd /tmp && R06F_EXTRA='[{}]' timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg3 -- python "$OLDPWD/profiles/scripts/r06f_cfg3_sweep.py" > "$OLDPWD/$O/3_rocprof.txt" 2>&1); echo "rocprof rc=$?" | tee -a $O/summary.txt
f=$(find /tmp/prof_cfg3 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python profiles/scripts/r06f_cfg3_sweep.py  (one 200 k-column export + two whole-genome exports of config 3, default settings)" > $O/3_kernel_stats.txt; head -60 "$f" >> $O/3_kernel_stats.txt; }
cut -c1-130 $O/3_kernel_stats.txt | head -16
