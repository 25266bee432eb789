#!/bin/bash
# r06: the whole GPU suite and the bench line, as the driver runs them
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06z
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest -q -m gpu tests -p no:cacheprovider --timeout 900 > $O/1_tests.txt 2>&1; echo "GPU suite rc=$?" | tee $O/summary.txt
tail -n 4 $O/1_tests.txt
timeout 1000 python bench.py > $O/2_bench.json 2> $O/2_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
grep "^\[bench" $O/2_bench.err | tail -4
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06z/2_bench.json").read().strip().splitlines()[-1])
h=d['columns']['hal2maf_full']
print("value", d['value'], "ms/step", d['ms_per_step'], "frac", d['roofline']['frac'], "traffic", d['roofline']['traffic'])
print("cfg3:", h['seconds'], h['runs_seconds'], "r4path:", h['by_round_4s_path']['seconds'], h.get('cpu_baseline',{}).get('parity_with_gpu'))
print("unique:", h['unique']['seconds'], h['unique']['by_the_column_walk']['seconds'], h['unique']['same_text'], "multi:", h['unique']['export_multi']['seconds'], h['unique']['export_multi']['by_the_column_walk']['seconds'])
print("end_to_end", d['end_to_end']['value'], d['end_to_end']['seconds'])
print("features:", {k:(v.get('value'), v.get('seconds')) for k,v in d['features'].items() if isinstance(v, dict)})
print("hal2maf 8M", d['columns']['hal2maf']['value'], "depth_wig", d['columns']['depth_wig'].get('seconds'), "cfg5 wig", d['cfg5']['wig']['seconds'], "cold", d['cold']['ms'])
print("cfg4", d['cfg4']['ms_per_step'], "wide", d['wide']['ms_per_step'], "walk", d['walk']['ms_per_step'])
PY
