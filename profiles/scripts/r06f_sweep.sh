#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06f
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python profiles/scripts/r06f_cfg3_sweep.py > $O/1_sweep.jsonl 2> $O/1_sweep.err; echo "sweep rc=$?" | tee $O/summary.txt
cat $O/1_sweep.jsonl | cut -c1-700
grep -E "^====|CPU seconds" $O/1_sweep.err | cut -c1-500
