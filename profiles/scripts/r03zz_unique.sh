#!/bin/bash
# hal2maf --unique (hal2mafMP.py's mode) through the flat state machine: parity of the column tools, then its time on 2 M columns
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03zz
mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_maxrefgap.py tests/test_gpu_columns.py tests/test_gpu_realdata.py -q -x > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
B=hal_amd/_build
$B/hgxRandGen --minGenomes 2 --maxGenomes 10 --meanDegree 1.5 --minSegmentLength 50 --maxSegmentLength 200 --minSegments 70000 --maxSegments 140000 --maxBranchLength 3 --seed 2 /tmp/a01.hgx 2>/dev/null
HGX_MAF_TIMING=1 $B/hal2maf --refGenome Genome_9 --noAncestors --unique --length 2000000 /tmp/a01.hgx /tmp/o2.maf > $O/unique_2M.log 2>&1
HGX_MAF_TIMING=1 $B/hal2maf --refGenome Genome_9 --noAncestors --length 2000000 /tmp/a01.hgx /tmp/o3.maf > $O/plain_2M.log 2>&1
tail -n 3 $O/tests.log | cut -c1-200; cat $O/unique_2M.log $O/plain_2M.log | cut -c1-300
