#!/bin/bash
# r06 call 3: the bench line as the driver runs it, by itself; then kernel stats (csv) of the config-3 leg.
#   gpurun --timeout 1500 -- 'bash profiles/scripts/r06c_bench.sh'
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06c
mkdir -p $O
export TMPDIR=/tmp
timeout 1000 python bench.py > $O/1_bench.json 2> $O/1_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt



tail -n 25 $O/1_bench.err
tail -c 1500 $O/1_bench.json
