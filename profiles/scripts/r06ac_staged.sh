#!/bin/bash
# r06ac: k_lift_merged's records staged in LDS and stored as the span they are (HGX_LIFT_STAGED=1) against 40-byte stores 40 bytes apart
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06ac
mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 0 1; do
  export HGX_LIFT_STAGED=$v
  timeout 600 python bench.py --maf-full 0 --maf-columns 0 --columns 0 --features 0 --cpu-sample 0 --sustained-seconds 0 --text-path 0 > $O/bench_$v.json 2> $O/bench_$v.err; echo "staged=$v bench rc=$?" | tee -a $O/summary.txt
  python - <<PY
import json
d=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("staged=$v value %.3f G ms/step %.4f kernels %s one_plan %.4f" % (d['value']/1e9, d['ms_per_step'], d['kernels_ms_per_step'], d['one_plan']['ms_per_step']))
print("   cfg4", d['cfg4']['ms_per_step'], d['cfg4']['one_plan']['ms_per_step'], d['cfg4']['kernels_ms_per_step'], "wide", d['wide']['ms_per_step'])
PY
done
export HGX_LIFT_STAGED=1
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider --timeout 600 tests/test_gpu_liftover.py tests/test_gpu_configs.py tests/test_gpu_pipelined.py tests/test_gpu_wide.py \
   tests/test_gpu_limits.py tests/test_gpu_composed.py tests/test_gpu_altpaths.py tests/test_gpu_textpath.py tests/test_gpu_exchange.py tests/test_gpu_multiseq.py tests/test_gpu_coalescence.py \
   "tests/test_gpu_zz_round5.py::test_config4_full_size_sample_vs_oracle" > $O/1_tests.txt 2>&1; echo "tests (staged) rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/1_tests.txt
