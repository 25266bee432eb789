#!/bin/bash
# round 3, first GPU call: the whole GPU suite on the int64 single-pass build, then the default bench line (new legs: wide, cfg4, cfg5,
# full config 3, cold phases)
O=gpurun_out/r03a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -3 $O/tests.log
tail -c 600 $O/bench.err
