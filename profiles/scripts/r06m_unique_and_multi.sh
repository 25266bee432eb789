#!/bin/bash
# r06: --unique and hgx_maf_export_multi after the keys-only stretches ship one column (bench.py's config-3 leg), the unique tests, kernel stats
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06m
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_gpu_unique.py tests/test_gpu_zz_round5.py -p no:cacheprovider --timeout 600 > $O/1_tests.txt 2>&1; echo "unique + round-5 tests rc=$?" | tee $O/summary.txt
tail -n 3 $O/1_tests.txt
HGX_MAF_TIMING=1 timeout 400 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/2_leg.json 2> $O/2_leg.err; echo "leg rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
h=json.loads(open("gpurun_out/r06m/2_leg.json").read().strip().splitlines()[-1])
print("cfg3", h["seconds"], h["runs_seconds"])
u=h["unique"]
print("unique", u["seconds"], u["runs_seconds"], "walk", u["by_the_column_walk"]["seconds"], u["same_text"], u["device_stage"])
print("multi", u["export_multi"]["seconds"], "walk", u["export_multi"]["by_the_column_walk"]["seconds"], u["export_multi"]["same_size"])
PY
grep "CPU seconds" $O/2_leg.err | tail -12 | cut -c1-400
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_u -- python "$OLDPWD/bench.py" --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > "$OLDPWD/$O/3_rocprof.txt" 2>&1); echo "rocprof rc=$?" | tee -a $O/summary.txt
f=$(find /tmp/prof_u -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0" > $O/3_kernel_stats.txt; head -60 "$f" >> $O/3_kernel_stats.txt; }
cut -c1-150 $O/3_kernel_stats.txt | head -26
