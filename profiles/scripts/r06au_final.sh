#!/bin/bash
# r06au (the last measurement pass of round 6): the library as built — PMC traffic (profiles/pmc_traffic.json), the whole GPU suite, the bench line, rocprofv3 of the timed form
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06au
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python profiles/scripts/r05_pmc.py /tmp/r06au_pmc > $O/1_pmc.txt 2>&1; echo "pmc rc=$?" | tee $O/summary.txt
cp /tmp/r06au_pmc/kernel_stats.txt $O/1_pmc_driver_kernel_stats.txt 2>/dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
grep -E "calibration|rotating|k_lift_classify|k_lift_merged|k_up_chain|k_sweep_up" $O/1_pmc.txt | cut -c1-250 | head -20
timeout 1500 python -m pytest -q -m gpu tests -p no:cacheprovider --timeout 900 > $O/2_tests.txt 2>&1; echo "GPU suite rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/2_tests.txt
L="--cfg4 0 --wide 0 --cpu-sample 0 --maf-full 0 --maf-columns 0 --columns 0 --text-path 0 --features 0 --sustained-seconds 0"
( cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06au_prof -- python $R/bench.py $L > /tmp/r06au_prof.log 2>&1 )
f=$(find /tmp/r06au_prof -name '*kernel_stats.csv' | head -1)
[ -z "$f" ] && tail -5 /tmp/r06au_prof.log
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $L   (two batches in flight, four rotating: the form of value)" > $O/3_kernel_stats_in_flight.txt; head -40 "$f" >> $O/3_kernel_stats_in_flight.txt; }
( cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06au_prof2 -- python $R/profiles/scripts/column_depth_timing.py > /tmp/r06au_prof2.log 2>&1 )
f=$(find /tmp/r06au_prof2 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python profiles/scripts/column_depth_timing.py   (cfg2 leaf x3, --countDupes x3, root, cfg5 leaf x2)" > $O/3_kernel_stats_depth.txt; head -40 "$f" >> $O/3_kernel_stats_depth.txt; }
timeout 1000 python bench.py > $O/4_bench.json 2> $O/4_bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06au/4_bench.json").read().strip().splitlines()[-1])
h=d['columns']['hal2maf_full']
print("value", d['value'], "ms/step", d['ms_per_step'], "frac", d['roofline']['frac'], "traffic", d['roofline']['traffic'])
print("cfg3:", h['seconds'], h['runs_seconds'], "r4path:", h['by_round_4s_path']['seconds'], h.get('cpu_baseline',{}).get('parity_with_gpu'))
print("unique:", h['unique']['seconds'], h['unique']['by_the_column_walk']['seconds'], h['unique']['same_text'], "multi:", h['unique']['export_multi']['seconds'], h['unique']['export_multi']['by_the_column_walk']['seconds'])
print("end_to_end", d['end_to_end']['value'], d['end_to_end']['seconds'])
print("features:", {k:(v.get('value'), v.get('seconds')) for k,v in d['features'].items() if isinstance(v, dict)})
print("hal2maf 8M", d['columns']['hal2maf']['value'], "depth kernel ms", d['columns'].get('kernel_ms'), "depth_wig", d['columns']['depth_wig'].get('seconds'), "cfg5", d['cfg5'].get('kernel_ms'), "cfg5 wig", d['cfg5']['wig']['seconds'], "cold", d['cold']['ms'])
print("cfg4", d['cfg4']['ms_per_step'], "wide", d['wide']['ms_per_step'], "walk", d['walk']['ms_per_step'], "one_plan", d['one_plan']['ms_per_step'], d['one_plan']['kernels_ms_per_step'])
PY
