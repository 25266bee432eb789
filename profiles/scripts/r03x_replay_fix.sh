#!/bin/bash
# the replay iterator with the paralogy cycles that go on after a walk is abandoned; --unique through it: column tests, the
# features soak, the general soak (MAF with --unique in half of its exports)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03x
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_columns.py tests/test_gpu_maxrefgap.py tests/test_gpu_altpaths.py tests/test_gpu_realdata.py tests/test_gpu_multiseq.py tests/test_gpu_cli.py tests/test_gpu_configs.py -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 300 python profiles/scripts/r03_features_soak.py 110 > $O/features_soak.log 2>&1
SOAK_SEED=9 timeout 200 python profiles/scripts/soak_parity.py 70 > $O/soak.log 2>&1
tail -n 3 $O/tests.log | cut -c1-300; tail -n 2 $O/features_soak.log | cut -c1-500; tail -n 1 $O/soak.log | cut -c1-300
