#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python profiles/scripts/r04l_text_timing.py 2>&1 | grep -v amdgpu | tail -6
timeout 600 python -m pytest tests/test_gpu_textpath.py tests/test_gpu_cli.py tests/test_gpu_columns.py -q 2>&1 | tail -2
python bench.py --steps 20 --cfg4 0 --wide 0 --cpu-sample 0 --rotating 0 --features 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['columns']; print('hal2maf 8M', c['hal2maf']['value']/1e6, 'full', c['hal2maf_full']['value']/1e6, c['hal2maf_full']['seconds'], 'end_to_end', d['end_to_end']['value']/1e6, d['end_to_end']['seconds'])"
