#!/bin/bash
# r06bd: the library of the round's last commit — smoke(), the GPU suite, the bench line
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06bd
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/0_smoke.txt 2>&1; echo "smoke rc=$? : $(tail -n 1 $O/0_smoke.txt | cut -c1-100)" | tee $O/summary.txt
timeout 1500 python -m pytest -q -m gpu tests -p no:cacheprovider --timeout 900 > $O/1_tests.txt 2>&1; echo "GPU suite rc=$? : $(tail -n 1 $O/1_tests.txt)" | tee -a $O/summary.txt
timeout 1000 python bench.py > $O/2_bench.json 2> $O/2_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06bd/2_bench.json").read().strip().splitlines()[-1])
h=d['columns']['hal2maf_full']
print("value %.3f G ms/step %.4f frac %.3f traffic %s" % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']))
print("cfg3 %.3f unique %.3f multi %s child %s cfg4 %.3f one_plan %.4f depth %.3f cfg5 %.2f wig %.4f %.4f e2e %.4f cold %.2f" % (h['seconds'], h['unique']['seconds'], h['unique'].get('export_multi',{}).get('seconds'), h.get('child_ended_with',{}).get('code'), d['cfg4']['ms_per_step'], d['one_plan']['ms_per_step'], d['columns']['kernel_ms'], d['cfg5']['kernel_ms'], d['columns']['depth_wig']['seconds'], d['cfg5']['wig']['seconds'], d['end_to_end']['seconds'], d['cold']['ms']))
print("parity:", d['cpu_baseline'].get('parity_with_gpu'), d['columns'].get('cpu_baseline',{}).get('parity_with_gpu'), h.get('cpu_baseline',{}).get('parity_with_gpu'), d['cfg5'].get('cpu_baseline',{}).get('parity_with_gpu'))
PY
