#!/bin/bash
# k_lift_classify with worker workgroups for the general intervals: parity, then the bench with and without them
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03k
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_maxrefgap.py --deselect tests/test_gpu_blockviz.py > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err
HGX_LIFT_WORKERS=0 timeout 600 python bench.py --steps 200 > $O/bench_noworkers.log 2> $O/bench_noworkers.err
grep -n "FAILED\|^E  " $O/tests.log | head -20 | cut -c1-300; tail -3 $O/tests.log
python - <<'PY'
import json
for f in ("bench", "bench_noworkers"):
    for l in open("gpurun_out/r03k/%s.log" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f, d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["one_plan"]["ms_per_step"], d["counts_per_step"]["general_queries"], d.get("cfg4", {}).get("ms_per_step"))
PY
