#!/usr/bin/env python3
"""Issue-side counters of the single-pass kernels (one rocprofv3 --pmc pass per counter group, kernel-trace only):
what the wavefronts of k_lift_classify / k_lift_merged spend their cycles on.  usage: r02k_sq.py <outdir>"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
driver = [sys.executable, os.path.join(ROOT, "profiles", "scripts", "r02_pmc_driver.py"), "1.0", "1000000", "--no-columns"]
GROUPS = [["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"],
          ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"],
          ["SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
          ["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_INSTS_SMEM"]]
res = defaultdict(dict)
for gi, group in enumerate(GROUPS):
    d = os.path.join(out, "g%d" % gi)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + driver,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        print("# group %s failed (rc %d): %s" % (group, r.returncode, r.stdout[-400:]))
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]
            a = acc[(name, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    for (name, counter), (total, n) in acc.items():
        res[name][counter] = total / n
for name in sorted(res):
    if name.startswith("k_lift") or name.startswith("k_up") or name.startswith("k_finish") or name.startswith("k_down"):
        print(name, " ".join("%s=%.4g" % kv for kv in sorted(res[name].items())))
