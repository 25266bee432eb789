#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06ae
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider --timeout 600 tests/test_gpu_pipelined.py > $O/1_tests.txt 2>&1; echo "tests rc=$?" | tee $O/summary.txt
tail -n 5 $O/1_tests.txt
