#!/bin/bash
# r06u: k_lift_classify's scouts (workers that find their general intervals themselves) — the liftover suites, then the timed form with
# and without them; --noDupes through the tracks
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06u
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider --timeout 600 tests/test_gpu_liftover.py tests/test_gpu_configs.py tests/test_gpu_pipelined.py tests/test_gpu_wide.py \
   tests/test_gpu_limits.py tests/test_gpu_composed.py tests/test_gpu_altpaths.py tests/test_gpu_textpath.py tests/test_gpu_exchange.py tests/test_gpu_multiseq.py tests/test_gpu_coalescence.py \
   "tests/test_gpu_zz_round5.py::test_maf_tracks_no_dupes" "tests/test_gpu_zz_round5.py::test_config4_full_size_sample_vs_oracle" > $O/1_tests.txt 2>&1; echo "tests rc=$?" | tee $O/summary.txt
tail -n 5 $O/1_tests.txt
L="--cfg4 0 --wide 0 --cpu-sample 0 --maf-full 0 --maf-columns 0 --columns 0 --text-path 0 --features 0 --sustained-seconds 0"
for w in scouts list0 scouts2; do
  if [ $w = list0 ]; then export HGX_LIFT_SCOUT=0; else unset HGX_LIFT_SCOUT; fi
  timeout 300 python bench.py $L > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?" | tee -a $O/summary.txt
  python - <<PY
import json
d=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1])
print("$w: value %.3f G  ms/step %.4f  kernels %s  frac %.3f  cached %.3f G one_plan %.4f" % (d['value']/1e9, d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['frac'], d['cached']['value']/1e9, d['one_plan']['ms_per_step']))
PY
done
timeout 600 python bench.py --maf-full 0 --maf-columns 0 --columns 0 --features 0 --cpu-sample 0 --sustained-seconds 0 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
d=json.loads(open("$O/bench_cfg4.json").read().strip().splitlines()[-1])
print("cfg4", d['cfg4']['ms_per_step'], "wide", d['wide']['ms_per_step'], "end_to_end", d['end_to_end']['seconds'], "cold", d['cold']['ms'])
PY
