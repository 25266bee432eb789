# rocprofv3 kernel stats of the bench with one batch at a time: the launches do not overlap, so the averages are the
# kernels' own durations (the ones bench.py's roofline objects use, measured there with HIP events)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02ab_prof -- python "$GRAFT_REPO_ROOT/bench.py" --in-flight 1 --columns 0 --maf-columns 0 --text-path 0 --cpu-sample 0 > /tmp/r02ab_bench.log 2>&1 )
f=$(find /tmp/r02ab_prof -name '*kernel_stats.csv' | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --in-flight 1 --columns 0 --maf-columns 0 --text-path 0 --cpu-sample 0" > gpurun_out/r02ab_kernel_stats_one_plan.txt
head -30 "$f" >> gpurun_out/r02ab_kernel_stats_one_plan.txt
grep '^{"metric"' /tmp/r02ab_bench.log | tail -1 > gpurun_out/r02ab_bench_one_plan.log
grep "k_lift" gpurun_out/r02ab_kernel_stats_one_plan.txt | cut -c1-40,300-420
python - <<PY
import json
d=json.loads(open("gpurun_out/r02ab_bench_one_plan.log").read()); print(d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["config"]["batches_in_flight"])
PY
