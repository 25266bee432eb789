#!/bin/bash
# r06y: the kernels of a depth run (cfg2 leaf + root, cfg5 leaf) by rocprofv3
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06y
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06y_p -- python $R/profiles/scripts/column_depth_timing.py > /tmp/r06y_p.log 2>&1 )
f=$(find /tmp/r06y_p -name '*kernel_stats.csv' | head -1)
[ -z "$f" ] && tail -5 /tmp/r06y_p.log
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python profiles/scripts/column_depth_timing.py" > $O/kernel_stats_depth.txt; head -40 "$f" >> $O/kernel_stats_depth.txt; }
t=$(find /tmp/r06y_p -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python - "$t" > $O/kernel_trace_depth.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    n = r["Kernel_Name"]
    if "sweep" not in n and "wig" not in n: 
        prev = int(r["End_Timestamp"]); continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = n.split("(")[0].replace("void hgx::", "")
    print("%-70s %9.1f us   gap %7.1f us  grid %s" % (short[:70], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, r.get("Grid_Size", "")))
    prev = e
PY
head -80 $O/kernel_trace_depth.txt
