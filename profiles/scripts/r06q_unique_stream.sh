#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06q
mkdir -p $O
export TMPDIR=/tmp
HGX_MAF_STREAM_DEBUG=1 timeout 900 python -m pytest -q -m gpu tests/test_gpu_columns.py tests/test_gpu_unique.py tests/test_gpu_zz_round5.py tests/test_gpu_maxrefgap.py tests/test_gpu_cli.py -p no:cacheprovider --timeout 600 > $O/1_tests.txt 2>&1; echo "column tests rc=$?" | tee $O/summary.txt
grep -E "hgx\]|passed|failed" $O/1_tests.txt | head -20
HGX_MAF_TIMING=1 timeout 400 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/2_leg.json 2> $O/2_leg.err; echo "leg rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
h=json.loads(open("gpurun_out/r06q/2_leg.json").read().strip().splitlines()[-1])
u=h["unique"]
print("cfg3", h["runs_seconds"], "unique", u["runs_seconds"], "walk", u["by_the_column_walk"]["seconds"], u["same_text"], u["device_stage"])
print("multi", u["export_multi"]["seconds"], "walk", u["export_multi"]["by_the_column_walk"]["seconds"], u["export_multi"]["same_size"], u["export_multi"]["tracks"])
PY
