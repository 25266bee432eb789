#!/bin/bash
# r06 call 1: the round-5 GPU test file alone, WITHOUT -x (VERDICT r05 "next" 1c), each test under its own limit.
#   gpurun --timeout 1500 -- 'bash profiles/scripts/r06a_tests_round5.sh'
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06a
mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showmeminfo vram > $O/0_smi.txt 2>&1
nproc >> $O/0_smi.txt
timeout 1300 python -m pytest -q -m gpu tests/test_gpu_zz_round5.py -p no:cacheprovider --timeout 600 --durations=20 > $O/1_round5_tests.txt 2>&1
echo "round-5 tests rc=$?" | tee $O/summary.txt
tail -n 40 $O/1_round5_tests.txt
