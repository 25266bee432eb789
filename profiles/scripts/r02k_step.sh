#!/bin/bash
# step time of the cfg2 batch under the environment variants given as arguments ("A=1,B=2" per variant; "-" = default)
mkdir -p gpurun_out
for v in "$@"; do
  ( [ "$v" != "-" ] && export $(echo "$v" | tr ',' ' '); timeout 300 python profiles/scripts/r02_merged_step.py 1.0 1000000 cfg2 "$v" ) 2>&1 | grep -v "^$\|amdgpu.ids"
done > gpurun_out/r02k_step.txt
cat gpurun_out/r02k_step.txt
