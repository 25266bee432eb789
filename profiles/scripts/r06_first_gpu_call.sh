#!/bin/bash
# The first GPU call of the round after round 5 (whose code never met a GPU: profiles/r05_notes.md): the round-5 tests, the bench line,
# kernel times of the new device stage, in one gpurun call with its own limits so that a hang costs minutes, not a strike.
#   gpurun --timeout 1500 -- 'bash profiles/scripts/r06_first_gpu_call.sh'
# Everything lands under gpurun_out/r06_first/.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06_first
mkdir -p $O
export TMPDIR=/tmp
# 1. the cheapest sign of life of the new device code: the reference's own goldens through the per-base tracks
timeout 300 python -m pytest -x -q -m gpu "tests/test_gpu_zz_round5.py::test_maf_tracks_reference_goldens" > $O/1_goldens.txt 2>&1; echo "goldens rc=$?" | tee -a $O/summary.txt
# 2. the round-5 file (tracks, --unique, the walk over slices at full size, wig chunks, config 4's oracle sample, the writers, the mp tools)
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_zz_round5.py --durations=15 > $O/2_round5_tests.txt 2>&1; echo "round-5 tests rc=$?" | tee -a $O/summary.txt
# 3. the line
timeout 600 python bench.py > $O/3_bench.json 2> $O/3_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
# 4. config 3 with the timing lines of the export (who waits for whom) and the kernels of the device stage
HGX_MAF_TIMING=1 timeout 300 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/4_cfg3_leg.json 2> $O/4_cfg3_timing.txt; echo "cfg3 leg rc=$?" | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/prof_cfg3" -- python "$OLDPWD/bench.py" --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > "$OLDPWD/$O/5_rocprof.txt" 2>&1); echo "rocprof rc=$?" | tee -a $O/summary.txt
find $O/prof_cfg3 -name "*kernel_stats*" -exec cp {} $O/5_kernel_stats.csv \; 2>/dev/null
rm -rf $O/prof_cfg3
tail -n 3 $O/2_round5_tests.txt
tail -c 600 $O/3_bench.json
