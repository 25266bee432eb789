#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06p
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_gpu_columns.py tests/test_gpu_unique.py tests/test_gpu_zz_round5.py -p no:cacheprovider --timeout 600 > $O/1_tests.txt 2>&1; echo "column tests rc=$?" | tee $O/summary.txt
tail -n 3 $O/1_tests.txt
for per in 1 2 3; do
HGX_MAF_MULTI_PER_HANDLE=$per timeout 400 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/2_leg_$per.json 2> $O/2_leg_$per.err; echo "leg per-handle $per rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
h=json.loads(open("gpurun_out/r06p/2_leg_$per.json").read().strip().splitlines()[-1])
u=h["unique"]
print("per handle $per: cfg3", h["runs_seconds"], "unique", u["runs_seconds"], "multi", u["export_multi"]["seconds"], "walk", u["export_multi"]["by_the_column_walk"]["seconds"], u["export_multi"]["same_size"])
PY
done
export R06F_EXTRA='[{"HGX_MAF_WALK_THREADS":"16"},{"HGX_MAF_WALK_THREADS":"24"},{"HGX_MAF_WALK_THREADS":"32"},{"HGX_MAF_WALK_THREADS":"24","HGX_MAF_RENDERS_IN_FLIGHT":"6"}]'
timeout 600 python profiles/scripts/r06f_cfg3_sweep.py > $O/3_sweep.jsonl 2> $O/3_sweep.err; echo "sweep rc=$?" | tee -a $O/summary.txt
cat $O/3_sweep.jsonl | cut -c1-330
