#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04j
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_blockviz.py tests/test_gpu_blockmap.py -q 2>&1 | tail -2
HGX_COMPOSED_UP=1 timeout 600 python -m pytest tests/test_gpu_blockviz.py tests/test_gpu_blockmap.py -q 2>&1 | tail -2
HGX_COMPOSED_AFTER=0.001 timeout 600 python -m pytest tests/test_gpu_blockviz.py tests/test_gpu_blockmap.py -q 2>&1 | tail -2
timeout 400 python profiles/scripts/r03_features_soak.py 90 > $O/features_soak.log 2>&1; tail -2 $O/features_soak.log | cut -c1-300
python bench.py --steps 20 --cfg4 0 --wide 0 --cpu-sample 0 --maf-full 0 --columns 0 --text-path 0 --rotating 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(d['features']['blocks_in_target_range'])[300:1200])"
