#!/bin/bash
# the committed library (.ab_old, a worktree of HEAD) and the working tree's on the same GPU, twice each, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p $O
for i in 1 2; do
  ( cd .ab_old && timeout 300 python bench.py --maf-full 0 --wide 0 --cfg4 0 --text-path 0 --cpu-sample 0 --cpu-all-cores 0 --sustained-seconds 0 > $O/old_$i.log 2> $O/old_$i.err )
  timeout 300 python bench.py --maf-full 0 --wide 0 --cfg4 0 --text-path 0 --cpu-sample 0 --cpu-all-cores 0 --sustained-seconds 0 > $O/new_$i.log 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03m/*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"] / 1e9, 3), round(d["ms_per_step"], 4), d["kernels_ms_per_step"], round(d["one_plan"]["ms_per_step"], 4), d["one_plan"].get("kernels_ms_per_step"))
PY
for f in $O/*.err; do tail -2 $f | cut -c1-200; done
