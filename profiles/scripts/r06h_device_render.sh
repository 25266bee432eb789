#!/bin/bash
# r06: hal2maf's text rendered on the device — the column suites (goldens, oracle comparisons) with it, then config 3 with it and without
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06h
mkdir -p $O
export TMPDIR=/tmp
HGX_MAF_DEVICE_RENDER_MIN_BLOCKS=1 timeout 900 python -m pytest -q -m gpu tests/test_gpu_columns.py tests/test_gpu_unique.py tests/test_gpu_zz_round5.py tests/test_gpu_maxrefgap.py -p no:cacheprovider --timeout 600 > $O/1_tests.txt 2>&1; echo "column tests (every batch rendered on the device) rc=$?" | tee $O/summary.txt
tail -n 5 $O/1_tests.txt
export R06F_EXTRA='[{"HGX_MAF_SLICED":"0"},{"HGX_MAF_SLICED":"1"},{"HGX_MAF_SLICED":"1","HGX_MAF_RENDERS_IN_FLIGHT":"4"},{"HGX_MAF_SLICED":"1","HGX_MAF_CHUNK":"1048576"},{"HGX_MAF_SLICED":"1","HGX_MAF_CHUNK":"1048576","HGX_MAF_RENDERS_IN_FLIGHT":"4"},{"HGX_MAF_SLICED":"0","HGX_MAF_DEVICE_RENDER":"0"},{"HGX_MAF_SLICED":"1","HGX_MAF_DEVICE_RENDER":"0"}]'
timeout 600 python profiles/scripts/r06f_cfg3_sweep.py > $O/2_sweep.jsonl 2> $O/2_sweep.err; echo "sweep rc=$?" | tee -a $O/summary.txt
cat $O/2_sweep.jsonl | cut -c1-600
grep -E "^====|CPU seconds" $O/2_sweep.err | cut -c1-460
