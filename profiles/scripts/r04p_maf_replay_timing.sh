#!/bin/bash
# the host side of hal2maf over recorded device batches (10 %-scale config-3 alignment; make hostprof-lib; HGX_MAF_REPLAY) on the
# GPU box's quiet cores (the recording of profiles/scripts/r04o_maf_dump.sh, kept beside the build: hal_amd/_build/maf_batches_r04o.bin)
cd "$GRAFT_REPO_ROOT" || exit 1
B=hal_amd/_build
$B/hgxRandGen --minGenomes 2 --maxGenomes 10 --meanDegree 1.5 --minSegmentLength 50 --maxSegmentLength 200 --minSegments 70000 --maxSegments 140000 --maxBranchLength 3 --seed 2 /tmp/a01.hgx 2>/dev/null
for i in 1 2 3; do
LD_PRELOAD=hal_amd/libhgx_hostprof.so HGX_MAF_REPLAY=hal_amd/_build/maf_batches_r04o.bin HGX_MAF_TIMING=1 $B/hal2maf --device -1 --refGenome Genome_9 --noAncestors /tmp/a01.hgx /tmp/o.maf 2>&1 | grep -E "state machine|Mticks" | cut -c1-200
done
md5sum /tmp/o.maf
