#!/bin/bash
# r06w: scouts that take exactly the intervals the tile's scan calls general for a flag or their length: liftover suites, then the diagnosis
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06w
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python profiles/scripts/r06v_scout_diag.py 100 > $O/diag.txt 2>&1; tail -4 $O/diag.txt
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider --timeout 600 tests/test_gpu_liftover.py tests/test_gpu_configs.py tests/test_gpu_pipelined.py tests/test_gpu_wide.py \
   tests/test_gpu_limits.py tests/test_gpu_composed.py tests/test_gpu_altpaths.py tests/test_gpu_textpath.py tests/test_gpu_exchange.py tests/test_gpu_multiseq.py tests/test_gpu_coalescence.py \
   "tests/test_gpu_zz_round5.py::test_config4_full_size_sample_vs_oracle" tests/test_gpu_columns.py > $O/1_tests.txt 2>&1; echo "tests rc=$?" | tee $O/summary.txt
tail -n 5 $O/1_tests.txt
