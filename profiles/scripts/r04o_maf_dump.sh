#!/bin/bash
# records the device's batches of hal2maf --refGenome Genome_9 --noAncestors over the 10 %-scale config-3 alignment (for replaying
# the host state machine on a machine without a GPU: make hostprof-lib, HGX_MAF_REPLAY)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04o
mkdir -p $O
B=hal_amd/_build
$B/hgxRandGen --minGenomes 2 --maxGenomes 10 --meanDegree 1.5 --minSegmentLength 50 --maxSegmentLength 200 --minSegments 70000 --maxSegments 140000 --maxBranchLength 3 --seed 2 /tmp/a01.hgx 2>/dev/null
md5sum /tmp/a01.hgx > $O/log.txt
LD_PRELOAD=hal_amd/libhgx_hostprof.so HGX_MAF_DUMP=$O/maf_batches.bin HGX_MAF_TIMING=1 $B/hal2maf --refGenome Genome_9 --noAncestors /tmp/a01.hgx /tmp/o.maf >> $O/log.txt 2>&1
md5sum /tmp/o.maf >> $O/log.txt; ls -la /tmp/o.maf $O/maf_batches.bin >> $O/log.txt
cat $O/log.txt
