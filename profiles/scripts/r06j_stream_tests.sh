#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06j
mkdir -p $O
export TMPDIR=/tmp
HGX_MAF_STREAM_DEBUG=1 timeout 900 python -m pytest -q -s -m gpu tests/test_gpu_columns.py tests/test_gpu_unique.py tests/test_gpu_zz_round5.py tests/test_gpu_maxrefgap.py -p no:cacheprovider --timeout 600 > $O/1_tests.txt 2>&1; echo "column tests rc=$?" | tee $O/summary.txt
grep -E "hgx\]|passed|failed|Abort|fault" $O/1_tests.txt | head -30
tail -n 3 $O/1_tests.txt
