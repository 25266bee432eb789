#!/bin/bash
# records the device batches of the exports of profiles/scripts/maf_replay_cases.py (and checks them against the oracle on the way)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i
mkdir -p $O
HGX_LIB_PATH=hal_amd/libhgx_hostprof.so timeout 600 python profiles/scripts/maf_replay_cases.py dump $O/cases.bin > $O/log.txt 2>&1
echo "rc=$?" >> $O/log.txt; ls -la $O >> $O/log.txt; tail -5 $O/log.txt
