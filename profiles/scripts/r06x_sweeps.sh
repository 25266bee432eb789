#!/bin/bash
# r06x: the pipelined byte sweep (k_sweep_up_bytes) against the plain one, grids, and the kernels of a depth run
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06x
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in "words1 HGX_SWEEP_WORDS=1" "words0 HGX_SWEEP_WORDS=0" "words1again HGX_SWEEP_WORDS=1"; do
  set -- $v
  env $2 timeout 300 python profiles/scripts/column_depth_timing.py > $O/depth_$1.txt 2>&1; echo "== $1"; grep "depth" $O/depth_$1.txt | grep -v gen | cut -c1-120
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06x_p -- python $R/profiles/scripts/column_depth_timing.py > /tmp/r06x_p.log 2>&1 )
f=$(find /tmp/r06x_p -name '*kernel_stats.csv' | head -1)
[ -z "$f" ] && tail -5 /tmp/r06x_p.log
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python profiles/scripts/column_depth_timing.py" > $O/kernel_stats_depth.txt; head -40 "$f" >> $O/kernel_stats_depth.txt; }
cut -c1-90,200- $O/kernel_stats_depth.txt | grep -i "sweep" | sed 's/(.*)//' | head -30
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider --timeout 600 tests/test_gpu_columns.py tests/test_gpu_limits.py "tests/test_gpu_configs.py::test_config5_full_size_depth_properties" > $O/1_tests.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/1_tests.txt
