#!/bin/bash
# r06: HBM traffic per kernel by PMC passes (profiles/pmc_traffic.json for the library as built), then hgx_maf_export_multi again
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06n
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python profiles/scripts/r05_pmc.py /tmp/r06n_pmc > $O/1_pmc.txt 2>&1; echo "pmc rc=$?" | tee $O/summary.txt
cp /tmp/r06n_pmc/kernel_stats.txt $O/1_pmc_driver_kernel_stats.txt 2>/dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
grep -E "calibration|rotating|k_lift_classify|k_lift_merged|k_up_chain|k_sweep_up" $O/1_pmc.txt | cut -c1-250 | head -20
HGX_MAF_TIMING=1 timeout 400 python bench.py --leg hal2maf_full --scale 1.0 --cpu-sample 0 --cpu-columns 0 --cpu-all-cores 0 > $O/2_leg.json 2> $O/2_leg.err; echo "leg rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
h=json.loads(open("gpurun_out/r06n/2_leg.json").read().strip().splitlines()[-1])
print("cfg3", h["seconds"], h["runs_seconds"])
u=h["unique"]
print("unique", u["seconds"], u["runs_seconds"], "walk", u["by_the_column_walk"]["seconds"], u["same_text"])
print("multi", u["export_multi"]["seconds"], "walk", u["export_multi"]["by_the_column_walk"]["seconds"], u["export_multi"]["same_size"])
PY
