#!/bin/bash
# r06 call 5: the bench line as the driver runs it (the oracle pools bounded by the cgroup's memory), with the cgroup's CPU and memory counters around it.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06e
mkdir -p $O
export TMPDIR=/tmp
{ echo cpu.max; cat /sys/fs/cgroup/cpu.max; echo cpu.stat; cat /sys/fs/cgroup/cpu.stat; echo memory.events; cat /sys/fs/cgroup/memory.events; echo cpuset; cat /sys/fs/cgroup/cpuset.cpus.effective; } > $O/0_cgroup_before.txt 2>&1
timeout 1000 python bench.py > $O/1_bench.json 2> $O/1_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
{ echo cpu.stat; cat /sys/fs/cgroup/cpu.stat; echo memory.events; cat /sys/fs/cgroup/memory.events; echo memory.peak; cat /sys/fs/cgroup/memory.peak; } > $O/2_cgroup_after.txt 2>&1
grep "^\[bench" $O/1_bench.err | tail -40
cat $O/0_cgroup_before.txt $O/2_cgroup_after.txt
tail -c 300 $O/1_bench.json
