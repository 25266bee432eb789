#!/bin/bash
# soaks of this round's new paths: worker workgroups (int32 and int64 tables), then the general parity soak (liftover, MAF — the
# log-based state machine —, depth) with the table from the first batch on
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03p
mkdir -p $O
timeout 200 python profiles/scripts/r03_workers_soak.py 50 > $O/workers_soak.log 2>&1
HGX_COMPOSED_UP=1 timeout 400 python profiles/scripts/soak_parity.py 200 > $O/soak_merged.log 2>&1
SOAK_SEED=2 timeout 300 python profiles/scripts/soak_parity.py 120 > $O/soak_default.log 2>&1
timeout 300 python -m pytest tests/test_gpu_columns.py -q -k "block_length_breaks" 2>&1 | tail -2
for f in $O/*.log; do tail -n 2 $f | cut -c1-400; done
