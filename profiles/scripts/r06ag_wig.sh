#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06ag
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 300 python profiles/scripts/r06ag_wig.py > $O/wig.txt 2>&1; cat $O/wig.txt | tail -8
( cd /tmp && PYTHONPATH=$R timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06ag_prof -- python $R/profiles/scripts/r06ag_wig.py > /tmp/r06ag_prof.log 2>&1 )
f=$(find /tmp/r06ag_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python profiles/scripts/r06ag_wig.py" > $O/kernel_stats_wig.txt; head -24 "$f" >> $O/kernel_stats_wig.txt; }
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r06ag/kernel_stats_wig.txt').read().splitlines()[1:]))
for r in rows[1:12]:
    print(r[0].split("(")[0][:50].ljust(52), r[1].rjust(6), "total %8.1f ms" % (int(r[2])/1e6), "avg %8.1f us" % (float(r[3])/1e3), r[4])
PY
