#!/usr/bin/env python3
"""Run bench.py under rocprofv3 once per counter group (counters only, no tracing domains) and print the per-kernel
average of every counter.  usage: pmc_passes.py <outdir> [kernel-name-filter] [-- bench args]"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

GROUPS = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"],
    ["SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
]

out = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
bench_args = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--columns", "0"]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for gi, g in enumerate(GROUPS):
    d = os.path.join(out, "g%02d" % gi)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g + ["--output-format", "csv", "-d", d, "--", sys.executable, "bench.py"] + bench_args
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        print("# group %s failed (rc %d): %s" % (g, r.returncode, r.stdout[-300:].replace("\n", " | ")))
        continue
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if filt and filt not in k:
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("    %-40s avg/launch %18.1f   launches %d" % (c, s / n, n))
