#!/bin/bash
# the round's last soaks, on the sources the final pass (r04z) measured: the general parity soak (liftover through the text path
# with its packed records, hal2maf with and without --unique, depth) with the table from the first batch on and by the default
# policy, and the soak of the round-3 entry points (--maxRefGap, --global, --printTree, halGetBlocksInTargetRange)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04zz
mkdir -p $O
HGX_COMPOSED_UP=1 SOAK_SEED=21 timeout 260 python profiles/scripts/soak_parity.py 150 > $O/soak_merged.log 2>&1
SOAK_SEED=22 timeout 260 python profiles/scripts/soak_parity.py 130 > $O/soak_default.log 2>&1
SOAK_SEED=23 timeout 200 python profiles/scripts/r03_features_soak.py 100 > $O/soak_features.log 2>&1
for f in $O/*.log; do echo "== $f"; tail -n 2 $f | cut -c1-400; done
