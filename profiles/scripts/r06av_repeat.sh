#!/bin/bash
# r06av: the final library again and again — the GPU suite three times, the bench line twice, export_multi 4 x 6 passes: is anything left that fails now and then?
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06av
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 1500 python -m pytest -q -m gpu tests -p no:cacheprovider --timeout 900 > $O/tests_$i.txt 2>&1; echo "GPU suite $i rc=$? : $(tail -n 1 $O/tests_$i.txt)" | tee -a $O/summary.txt
done
for i in 1 2; do
  timeout 1000 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?" | tee -a $O/summary.txt
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06av/bench_$i.json").read().strip().splitlines()[-1])
h=d['columns']['hal2maf_full']
print("value %.3f G frac %.3f cfg3 %.3f unique %.3f multi %s child %s cfg4 %.3f depth %.3f cfg5 %.2f" % (d['value']/1e9, d['roofline']['frac'], h['seconds'], h['unique']['seconds'], h['unique'].get('export_multi',{}).get('seconds'), h.get('child_ended_with',{}).get('code'), d['cfg4']['ms_per_step'], d['columns']['kernel_ms'], d['cfg5']['kernel_ms']))
PY
done
for i in 1 2 3 4; do
  timeout 200 python profiles/scripts/r06al_multi.py 6 both > $O/multi_$i.txt 2>&1; echo "multi run $i rc=$? : $(tail -n 1 $O/multi_$i.txt | cut -c1-60)" | tee -a $O/summary.txt
done
