#!/bin/bash
# hal2maf with the log-based state machine (the rows made by the rendering threads): every MAF parity test, config 3 at full size
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03j
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_columns.py tests/test_gpu_maxrefgap.py tests/test_gpu_altpaths.py tests/test_gpu_cli.py tests/test_gpu_multiseq.py tests/test_gpu_realdata.py -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 900 python profiles/scripts/maf_cfg3_full.py > $O/maf_cfg3_full.log 2>&1
grep -n "FAILED\|^E  " $O/tests.log | head -30 | cut -c1-300; tail -3 $O/tests.log; cat $O/maf_cfg3_full.log | cut -c1-400
