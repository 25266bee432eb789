#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04g
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_liftover.py tests/test_gpu_composed.py tests/test_gpu_configs.py tests/test_gpu_blockmap.py tests/test_gpu_blockviz.py tests/test_gpu_coalescence.py tests/test_gpu_altpaths.py tests/test_gpu_multiseq.py -x -q > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -4 $O/tests.log
python profiles/scripts/r04f_cfg4_late.py 2>&1 | tail -2
HGX_LIB_PATH=$GRAFT_REPO_ROOT/hal_amd/libhgx_prof.so python profiles/scripts/r04f_cfg4_late.py 2>&1 | grep "finish profile" | tail -1
