#!/usr/bin/env python3
"""HBM traffic per kernel launch for bench.py's roofline objects, measured on the library as built:
  rocprofv3 --kernel-trace --stats                   -> <out>/kernel_stats.txt   (per-kernel durations)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE          -> bytes read  (own pass)
  rocprofv3 --kernel-trace --pmc WRITE_SIZE          -> bytes written (own pass)
over profiles/scripts/r05_pmc_driver.py, then writes profiles/pmc_traffic.json with the hash of every kernel's machine code in
hal_amd/libhgx.so (bench.kernel_code_sha16s), so that bench.py quotes a kernel's figure exactly while that kernel is the one measured.
FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts a wide coalesced streaming read at
half its bytes.  The factor is not assumed here: the driver launches, in the same passes, a 1 GiB device-to-device copy (1 GiB
must be read) and a gather of 16 Mi random 16-byte rows (one 128-byte line each), and
    stream_factor = bytes the copy must read / its FETCH_SIZE,   gather_factor = 128 B x rows / the gather's FETCH_SIZE
    write_factor  = bytes the copy must write / its WRITE_SIZE
are applied: `traffic` = FETCH_SIZE x stream_factor for the kernels that stream their inputs (STREAMING below: the record
compaction and the depth sweeps), FETCH_SIZE x gather_factor for the gather-bound kernels, and for the two single-pass kernels
- streamed per-interval inputs, gathered table records - the streamed bytes (STREAMED_PER_INTERVAL x intervals) are counted at
stream_factor and the rest at gather_factor; the raw values and the factors are kept in the file.
usage: r05_pmc.py <outdir> [driver args]"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STREAMING = {"k_compact_records", "k_depth_fill", "k_sweep_up", "k_sweep_down", "k_sweep_top", "k_sweep_out", "k_sum64", "k_scatter", "k_scatter_front"}
# kernels that stream their per-interval inputs and gather the rest: the streamed bytes (per interval) are the part counted at half
STREAMED_PER_INTERVAL = {"k_lift_classify": 16.0, "k_lift_merged": 33.0}
NQ = float(sys.argv[3]) if len(sys.argv) > 3 else 1e6
out = sys.argv[1]
driver = [sys.executable, os.path.join(ROOT, "profiles", "scripts", "r05_pmc_driver.py")] + sys.argv[2:]
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
        f.write("# rocprofv3 --kernel-trace --stats -- python profiles/scripts/r05_pmc_driver.py %s\n" % " ".join(sys.argv[2:]))
        f.write(open(stats[0]).read())
    agg = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(stats[0])):
        a = agg[short(row["Name"])]
        a[0] += float(row["TotalDurationNs"])
        a[1] += int(row["Calls"])
    dur = {k: v[0] / v[1] for k, v in agg.items()}
else:
    print("# kernel-trace pass failed:", r.stdout[-500:])
raw, peak = {}, {}
rot_raw = {}
for counter, form in (("FETCH_SIZE", "steady"), ("WRITE_SIZE", "steady"), ("FETCH_SIZE", "rotating"), ("WRITE_SIZE", "rotating")):
    d = os.path.join(out, counter + ("" if form == "steady" else "_rotating"))
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + driver +
                       ([] if form == "steady" else ["--form", "rotating"]), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        print("# %s pass failed (rc %d): %s" % (counter, r.returncode, r.stdout[-500:]))
        continue
    acc = defaultdict(lambda: [0.0, 0])
    top = defaultdict(float)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            top[short(row["Kernel_Name"])] = max(top[short(row["Kernel_Name"])], float(row["Counter_Value"]) * 1024.0)
    if form == "rotating":
        rot_raw[counter] = {k: v[0] / v[1] * 1024.0 for k, v in acc.items() if k.startswith("k_lift_")}
        continue
    raw[counter] = {k: v[0] / v[1] * 1024.0 for k, v in acc.items()}
    peak[counter] = dict(top)  # the largest launch of each kernel (the calibration copy among the library's small copies)


def largest(counter, part):
    return max([v for k, v in peak.get(counter, {}).items() if part in k] or [0.0])


GIB = float(1 << 30)
copy_fetch, copy_write, gather_fetch = largest("FETCH_SIZE", "copyBuffer"), largest("WRITE_SIZE", "copyBuffer"), largest("FETCH_SIZE", "gather_kernel")
calibration = {"copy_1GiB_fetch_raw": copy_fetch, "copy_1GiB_write_raw": copy_write, "gather_16Mi_rows_fetch_raw": gather_fetch,
               "stream_factor": GIB / copy_fetch if copy_fetch > 0.4 * GIB else None,
               "write_factor": GIB / copy_write if copy_write > 0.4 * GIB else None,
               # 16 Mi rows x 128 B lines + the 128 MiB of indices (streamed, seen at 1 / stream_factor)
               "gather_factor": None}
sf = calibration["stream_factor"] or 2.0
wf = calibration["write_factor"] or 1.0
if gather_fetch > 0:
    calibration["gather_factor"] = (float(1 << 24) * 128.0) / max(gather_fetch - float(1 << 27) / sf, 1.0)
gf = calibration["gather_factor"] or 1.0
print("# calibration:", json.dumps(calibration))
kernels, detail = {}, {}
for k in sorted(set(raw.get("FETCH_SIZE", {})) | set(raw.get("WRITE_SIZE", {}))):
    fe, wr = raw.get("FETCH_SIZE", {}).get(k, 0.0), raw.get("WRITE_SIZE", {}).get(k, 0.0)
    if k in STREAMING:
        read = fe * sf
    else:
        streamed = STREAMED_PER_INTERVAL.get(k, 0.0) * NQ  # seen by the counter as streamed / sf
        read = streamed + max(fe - streamed / sf, 0.0) * gf
    kernels[k] = read + wr * wf
    detail[k] = {"fetch_raw": fe, "write_raw": wr, "class": "streaming" if k in STREAMING else "gather",
                 "streamed_input_bytes": STREAMED_PER_INTERVAL.get(k, 0.0) * NQ, "avg_ns": dur.get(k)}
# the rotating form (bench.py: rotating): the same kernels when four batches, each with buffers of its own, take turns
rotating, rot_detail = {}, {}
for k in sorted(set(rot_raw.get("FETCH_SIZE", {})) | set(rot_raw.get("WRITE_SIZE", {}))):
    fe, wr = rot_raw.get("FETCH_SIZE", {}).get(k, 0.0), rot_raw.get("WRITE_SIZE", {}).get(k, 0.0)
    streamed = STREAMED_PER_INTERVAL.get(k, 0.0) * NQ
    rotating[k] = streamed + max(fe - streamed / sf, 0.0) * gf + wr * wf
    rot_detail[k] = {"fetch_raw": fe, "write_raw": wr}
sha = hashlib.sha256(open(os.path.join(ROOT, "hal_amd", "libhgx.so"), "rb").read()).hexdigest()[:16]
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha16, kernel_code_sha16s
# (round 5: every kernel's figure is tied to the hash of that kernel's machine code in the library measured — bench.py quotes a
# figure as long as the running library has the same code for that kernel, whatever else was rebuilt)
res = {"libhgx_sha16": sha, "kernel_sources_sha16": kernel_sources_sha16(), "kernel_code_sha16": kernel_code_sha16s(), "kernels": kernels, "detail": detail, "calibration": calibration,
       "rotating": rotating, "rotating_detail": rot_detail,
       "source": "profiles/scripts/r05_pmc.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (KiB x 1024) over the cfg2 batch, "
                 "scaled by the factors a 1 GiB copy and a 16 Mi-row gather of known traffic gave in the same passes (calibration); per launch"}
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
for k in sorted(rotating):
    print("rotating %-19s traffic %14.0f B   fetch_raw %14.0f  write_raw %14.0f" % (k, rotating[k], rot_detail[k]["fetch_raw"], rot_detail[k]["write_raw"]))
for k in sorted(kernels):
    print("%-28s traffic %14.0f B   fetch_raw %14.0f  write_raw %14.0f  avg %s ns" % (k, kernels[k], detail[k]["fetch_raw"], detail[k]["write_raw"], detail[k]["avg_ns"]))
