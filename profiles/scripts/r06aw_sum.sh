#!/bin/bash
# r06aw: the sums through k_sweep_up_words (--countDupes, hal2maf's tracks): depth timings, the column and tracks tests, config 3's leg
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06aw
mkdir -p $O
export TMPDIR=/tmp
for v in "words1 HGX_SWEEP_WORDS=1" "words0 HGX_SWEEP_WORDS=0"; do
  set -- $v
  env $2 timeout 300 python profiles/scripts/column_depth_timing.py > $O/depth_$1.txt 2>&1; echo "== $1"; grep "depth" $O/depth_$1.txt | grep -v gen | cut -c1-120
done
timeout 1500 python -m pytest -q -m gpu -p no:cacheprovider --timeout 900 tests/test_gpu_columns.py tests/test_gpu_unique.py tests/test_gpu_limits.py tests/test_gpu_zz_round5.py tests/test_gpu_configs.py > $O/1_tests.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/1_tests.txt
HGX_MAF_TIMING=1 timeout 400 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/leg.json 2> $O/leg.err; echo "leg rc=$?"
grep "tracks of genome" $O/leg.err | head -3
python - <<PY
import json
h=json.loads(open("gpurun_out/r06aw/leg.json").read().strip().splitlines()[-1])
u=h["unique"]
print("cfg3", h["seconds"], h["runs_seconds"], "tracks build ms", h['device_stage']['build_ms'], "unique", u["seconds"], "multi", u.get("export_multi",{}).get("seconds"))
PY
