#!/usr/bin/env python3
"""HBM traffic per kernel launch for bench.py's roofline objects, measured on the library as built:
  rocprofv3 --kernel-trace --stats                   -> <out>/kernel_stats.txt   (per-kernel durations)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE          -> bytes read  (own pass)
  rocprofv3 --kernel-trace --pmc WRITE_SIZE          -> bytes written (own pass)
over profiles/scripts/r02_pmc_driver.py, then writes profiles/pmc_traffic.json with the hash of hal_amd/libhgx.so, so that
bench.py only quotes the figures for the build they were measured on.
FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts a wide coalesced streaming read at
half its bytes; profiles/r01l_fetch_size_calibration.txt: a random 16-byte gather is counted in full.  `traffic` therefore
doubles FETCH_SIZE for the kernels that stream their inputs (STREAMING below), takes it as it is for the gather-bound walk
kernels, and for the two single-pass kernels — streamed per-interval inputs, gathered table records — adds the uncounted half of
the streamed bytes (STREAMED_PER_INTERVAL x intervals); the raw values are kept in the file.
usage: r02_pmc.py <outdir> [driver args]"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STREAMING = {"k_compact_records", "k_depth_fill"}
# kernels that stream their per-interval inputs and gather the rest: the streamed bytes (per interval) are the part counted at half
STREAMED_PER_INTERVAL = {"k_lift_classify": 16.0, "k_lift_merged": 33.0}
NQ = float(sys.argv[3]) if len(sys.argv) > 3 else 1e6
out = sys.argv[1]
driver = [sys.executable, os.path.join(ROOT, "profiles", "scripts", "r02_pmc_driver.py")] + sys.argv[2:]
os.makedirs(out, exist_ok=True)


def short(name):
    name = name.split("(")[0].replace("void ", "")
    name = name.split("<")[0]
    return name.split("::")[-1]


d = os.path.join(out, "trace")
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + driver, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT, text=True, timeout=900)
stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
dur = {}
if stats:
    with open(os.path.join(out, "kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python profiles/scripts/r02_pmc_driver.py %s\n" % " ".join(sys.argv[2:]))
        f.write(open(stats[0]).read())
    agg = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(stats[0])):
        a = agg[short(row["Name"])]
        a[0] += float(row["TotalDurationNs"])
        a[1] += int(row["Calls"])
    dur = {k: v[0] / v[1] for k, v in agg.items()}
else:
    print("# kernel-trace pass failed:", r.stdout[-500:])
raw = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out, counter)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + driver, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        print("# %s pass failed (rc %d): %s" % (counter, r.returncode, r.stdout[-500:]))
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    raw[counter] = {k: v[0] / v[1] * 1024.0 for k, v in acc.items()}
kernels, detail = {}, {}
for k in sorted(set(raw.get("FETCH_SIZE", {})) | set(raw.get("WRITE_SIZE", {}))):
    fe, wr = raw.get("FETCH_SIZE", {}).get(k, 0.0), raw.get("WRITE_SIZE", {}).get(k, 0.0)
    kernels[k] = (2.0 * fe if k in STREAMING else fe + 0.5 * STREAMED_PER_INTERVAL.get(k, 0.0) * NQ) + wr
    detail[k] = {"fetch_raw": fe, "write_raw": wr, "fetch_doubled": k in STREAMING,
                 "streamed_input_bytes_counted_at_half": STREAMED_PER_INTERVAL.get(k, 0.0) * NQ, "avg_ns": dur.get(k)}
sha = hashlib.sha256(open(os.path.join(ROOT, "hal_amd", "libhgx.so"), "rb").read()).hexdigest()[:16]
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha16
res = {"libhgx_sha16": sha, "kernel_sources_sha16": kernel_sources_sha16(), "kernels": kernels, "detail": detail,
       "source": "profiles/scripts/r02_pmc.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (KiB x 1024) over the cfg2 batch; "
                 "FETCH_SIZE doubled for the streaming kernels (MI355X_MICROARCH.md, gfx950), as counted for the gather-bound walk kernels "
                 "(profiles/r01l_fetch_size_calibration.txt); per launch"}
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for k in sorted(kernels):
    print("%-28s traffic %14.0f B   fetch_raw %14.0f  write_raw %14.0f  avg %s ns" % (k, kernels[k], detail[k]["fetch_raw"], detail[k]["write_raw"], detail[k]["avg_ns"]))
