#!/bin/bash
# whole GPU suite + the bench line of this build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03n
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
grep -n "FAILED\|^E  " $O/tests.log | head -20 | cut -c1-300; tail -3 $O/tests.log; tail -2 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r03n/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["one_plan"]["ms_per_step"], d["one_plan"].get("kernels_ms_per_step"))
        print("cfg4", d.get("cfg4", {}).get("ms_per_step"), "wide", d.get("wide", {}).get("ms_per_step"), d.get("wide", {}).get("one_plan"))
        print("maf", d["columns"].get("hal2maf"), d["columns"].get("hal2maf_full"))
        print("cold", d["cold"]["ms"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
