#!/bin/bash
# finish_wave with extractSegment on masks: parity (suite + soak against the oracle), then the bench's liftover legs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03s
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_liftover.py tests/test_gpu_composed.py tests/test_gpu_pipelined.py tests/test_gpu_wide.py tests/test_gpu_configs.py tests/test_gpu_multiseq.py tests/test_gpu_realdata.py tests/test_gpu_textpath.py tests/test_gpu_coalescence.py -q -x > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
HGX_COMPOSED_UP=1 SOAK_SEED=5 timeout 300 python profiles/scripts/soak_parity.py 120 > $O/soak_merged.log 2>&1
HGX_COMPOSED_UP=1 HGX_FORCE_WIDE=1 SOAK_SEED=6 timeout 200 python profiles/scripts/soak_parity.py 60 > $O/soak_merged_wide.log 2>&1
timeout 400 python bench.py --columns 0 --wide 0 --text-path 0 --cpu-sample 0 --cpu-all-cores 0 --sustained-seconds 0 > $O/bench.log 2> $O/bench.err
tail -3 $O/tests.log | cut -c1-300; tail -n 1 $O/soak_merged.log | cut -c1-300; tail -n 1 $O/soak_merged_wide.log | cut -c1-300
python - <<'PY'
import json
for l in open("gpurun_out/r03s/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["one_plan"]["ms_per_step"], d["one_plan"].get("kernels_ms_per_step"), d["cfg4"]["ms_per_step"], d["cfg4"]["kernels_ms_per_step"])
PY
