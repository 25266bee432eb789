#!/bin/bash
# first GPU pass of round 4: the --unique device path (new tests + every column test), then the bench with the new legs
# (rotating, features) without the long ones
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unique.py tests/test_gpu_columns.py tests/test_gpu_multiseq.py tests/test_gpu_realdata.py tests/test_gpu_wide.py tests/test_gpu_maxrefgap.py tests/test_gpu_cli.py -x -q -s > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py --steps 20 --cfg4 0 --wide 0 --cpu-sample 0 --maf-full 0 > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -15 $O/tests.log; tail -3 $O/bench.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04a/bench.log").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    print("rotating", json.dumps(d.get("rotating"))[:1500])
    print("features", json.dumps(d.get("features"))[:3000])
except Exception as e:
    print("no bench line:", e)
PY
