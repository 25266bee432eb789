#!/bin/bash
# round 3, third GPU call: blockViz parity (HIP path vs the reference's expected output and the oracle), block-map regression, bench with
# shared streams for the wide leg
O=gpurun_out/r03c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blockviz.py tests/test_gpu_blockmap.py tests/test_gpu_coalescence.py -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py --cfg4 0 --maf-full 0 --cpu-sample 0 > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -30 $O/tests.log
