#!/bin/bash
# r06af: the rendered rows stored eight characters at a time — the column suites with every batch rendered on the device, config 3
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06af
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
HGX_MAF_DEVICE_RENDER_MIN_BLOCKS=1 timeout 1200 python -m pytest -q -m gpu -p no:cacheprovider --timeout 900 tests/test_gpu_columns.py tests/test_gpu_unique.py tests/test_gpu_maxrefgap.py \
   "tests/test_gpu_zz_round5.py::test_maf_tracks_reference_goldens" "tests/test_gpu_zz_round5.py::test_maf_tracks_vs_walk_and_oracle" "tests/test_gpu_zz_round5.py::test_maf_stream_and_device_render_against_the_host_paths" \
   "tests/test_gpu_zz_round5.py::test_maf_tracks_at_full_size" tests/test_gpu_cli.py > $O/1_tests.txt 2>&1; echo "tests rc=$?" | tee $O/summary.txt
tail -n 4 $O/1_tests.txt
for i in 1 2; do
HGX_MAF_TIMING=1 timeout 400 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/2_leg$i.json 2> $O/2_leg$i.err; echo "leg rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
h=json.loads(open("gpurun_out/r06af/2_leg$i.json").read().strip().splitlines()[-1])
print("cfg3", h["seconds"], h["runs_seconds"], h['device_stage']['last_export']['seconds'])
u=h["unique"]
print("unique", u["seconds"], u["runs_seconds"], "multi", u["export_multi"]["seconds"])
PY
done
( cd /tmp && PYTHONPATH=$R timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06af_prof -- python $R/bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > /tmp/r06af_prof.log 2>&1 )
f=$(find /tmp/r06af_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --leg hal2maf_full   (config 3's leg: two plain exports, round 4's path, --unique, export_multi)" > $O/3_kernel_stats_cfg3_leg.txt; head -30 "$f" >> $O/3_kernel_stats_cfg3_leg.txt; }
grep "k_maf_render_rows\|k_maf_rows_ctl\|k_maf_heads_out" $O/3_kernel_stats_cfg3_leg.txt | cut -c1-40,200-300
