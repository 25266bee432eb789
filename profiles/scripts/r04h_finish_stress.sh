#!/bin/bash
# the finishing kernels after the round's changes (extractSegment on bit sets, rank sorts, 128-piece staging, k_finish_big in LDS): the
# whole GPU suite, then the liftover suites with every general interval forced through finish_query (HGX_FINISH_WAVE=0) and on the
# round-1 path (HGX_MERGED=0), then the randomised soak the same two ways
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04h
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "all rc=$?" >> $O/tests_all.log
T="tests/test_gpu_liftover.py tests/test_gpu_composed.py tests/test_gpu_multiseq.py tests/test_gpu_coalescence.py tests/test_gpu_blockmap.py tests/test_gpu_realdata.py"
HGX_FINISH_WAVE=0 timeout 900 python -m pytest $T -q > $O/tests_finish_wave0.log 2>&1; echo "rc=$?" >> $O/tests_finish_wave0.log
HGX_MERGED=0 timeout 900 python -m pytest $T -q > $O/tests_merged0.log 2>&1; echo "rc=$?" >> $O/tests_merged0.log
HGX_COMPOSED_UP=1 HGX_FINISH_WAVE=0 timeout 300 python profiles/scripts/soak_parity.py 150 > $O/soak_finish_wave0.log 2>&1
HGX_COMPOSED_UP=1 HGX_MERGED=0 SOAK_SEED=3 timeout 300 python profiles/scripts/soak_parity.py 120 > $O/soak_merged0.log 2>&1
HGX_COMPOSED_UP=1 SOAK_SEED=4 timeout 300 python profiles/scripts/soak_parity.py 120 > $O/soak_default.log 2>&1
for f in $O/*.log; do echo "== $f"; tail -n 2 $f | cut -c1-300; done
