#!/bin/bash
# records the device's batches of hal2maf --unique over the first 100 k columns of the 10 %-scale config-3 alignment (profiling the
# column-by-column host path on a machine without a GPU)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z
mkdir -p $O
B=hal_amd/_build
$B/hgxRandGen --minGenomes 2 --maxGenomes 10 --meanDegree 1.5 --minSegmentLength 50 --maxSegmentLength 200 --minSegments 70000 --maxSegments 140000 --maxBranchLength 3 --seed 2 /tmp/a01.hgx 2>/dev/null
LD_PRELOAD=hal_amd/libhgx_hostprof.so HGX_MAF_DUMP=$O/unique_batches.bin HGX_MAF_TIMING=1 $B/hal2maf --refGenome Genome_9 --noAncestors --unique --length 100000 /tmp/a01.hgx /tmp/o.maf > $O/log.txt 2>&1
( time $B/hal2maf --refGenome Genome_9 --noAncestors --unique --length 1000000 /tmp/a01.hgx /tmp/o2.maf ) >> $O/log.txt 2>&1
md5sum /tmp/o.maf >> $O/log.txt; ls -la /tmp/o.maf /tmp/o2.maf $O/unique_batches.bin >> $O/log.txt
cat $O/log.txt
