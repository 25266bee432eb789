#!/bin/bash
# depth sweeps moving 8-byte words per lane: parity of the column tools, then the depth legs of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_columns.py tests/test_gpu_multiseq.py tests/test_gpu_altpaths.py tests/test_gpu_realdata.py tests/test_gpu_configs.py -q -x > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py --maf-full 0 --wide 0 --text-path 0 --cpu-sample 0 --cpu-all-cores 0 --sustained-seconds 0 --maf-columns 0 > $O/bench.log 2> $O/bench.err
grep -n "FAILED\|^E  " $O/tests.log | head -20 | cut -c1-300; tail -3 $O/tests.log; tail -2 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r03o/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        c = d["columns"]
        print("cfg2 depth", c["kernel_ms"], c["value"], c["roofline"]["sweeps_own_frac"])
        print("cfg5", d.get("cfg5"))
PY
