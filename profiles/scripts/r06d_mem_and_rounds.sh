#!/bin/bash
# r06 call 4: what the box has (memory, cgroup), config 3 with the walk's per-round timing, then the bench line WITHOUT the oracle pools.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06d
mkdir -p $O
export TMPDIR=/tmp
{ free -g; echo; cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory.current 2>&1; echo; df -h /tmp /dev/shm; echo; nproc; lscpu | head -30; echo; numactl -H 2>&1 | head -20; echo; ulimit -a; which perf gdb strace; } > $O/0_box.txt 2>&1
HGX_MAF_TIMING=1 timeout 300 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/1_cfg3_leg.json 2> $O/1_cfg3_timing.txt; echo "cfg3 leg rc=$?" | tee -a $O/summary.txt
grep -E "round [0-9]+: [0-9]+ walk|comparing|the walk over" $O/1_cfg3_timing.txt | head -60
timeout 900 python bench.py --cpu-all-cores 0 > $O/2_bench.json 2> $O/2_bench.err; echo "bench (no pools) rc=$?" | tee -a $O/summary.txt
grep "^\[bench" $O/2_bench.err | tail -40
tail -c 600 $O/2_bench.json
