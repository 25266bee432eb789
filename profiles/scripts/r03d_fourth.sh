#!/bin/bash
# round 3, fourth GPU call: hal2maf --maxRefGap through the HIP path (gap kernels + host replay) against the oracle; whole suite
O=gpurun_out/r03d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_maxrefgap.py -q > $O/gap_tests.log 2>&1
echo "pytest rc=$?" >> $O/gap_tests.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_maxrefgap.py > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -40 $O/gap_tests.log | cut -c1-400; tail -5 $O/tests.log
