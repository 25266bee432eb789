#!/bin/bash
# r06: the timed form (four batches rotating, two in flight) with the general intervals found up front and finished by worker workgroups
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06s
mkdir -p $O
export TMPDIR=/tmp
L="--cfg4 0 --wide 0 --cpu-sample 0 --maf-full 0 --maf-columns 0 --columns 0 --text-path 0 --features 0 --sustained-seconds 0"
for w in default 400 800; do
  if [ $w = default ]; then unset HGX_LIFT_WORKERS; else export HGX_LIFT_WORKERS=$w; fi
  timeout 300 python bench.py $L > $O/bench_$w.json 2> $O/bench_$w.err; echo "workers $w rc=$?" | tee -a $O/summary.txt
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06s/bench_$w.json").read().strip().splitlines()[-1])
print("workers $w: value %.3f G  ms/step %.4f  kernels %s  frac %.3f  cached %.3f G one_plan %.4f" % (d['value']/1e9, d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['frac'], d['cached']['value']/1e9, d['one_plan']['ms_per_step']))
PY
done
