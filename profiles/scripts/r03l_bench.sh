#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03l
mkdir -p $O
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err
echo "rc=$?" >> $O/bench.err
tail -5 $O/bench.err | cut -c1-400
python - <<'PY'
import json
for l in open("gpurun_out/r03l/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["one_plan"]["ms_per_step"], d["counts_per_step"]["general_queries"], d.get("cfg4", {}).get("ms_per_step"), d.get("wide", {}))
PY
