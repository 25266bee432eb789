#!/bin/bash
# round 3, fifth GPU call: --unique fix, --global, column tools over device clones, maxRefGap; whole suite
O=gpurun_out/r03e
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
grep -n "FAILED\|^E  " $O/tests.log | head -40 | cut -c1-300; tail -4 $O/tests.log
