#!/usr/bin/env python3
"""Ties the PMC traffic of profiles/pmc_traffic.json — measured in round 4 on the library of commit 3e8085e (libhgx_sha16 in the file) —
to the machine code of each kernel, so that bench.py can go on quoting a kernel's figure for exactly as long as that kernel's code is
the measured one (bench.kernel_code_sha16s; round 4's file-level hashes voided every figure at the first host-side commit).
No GPU is needed: the measured commit is rebuilt in a scratch directory (hipcc cross-compiles), the rebuilt libhgx.so must have the
sha256 the PMC run recorded (it does: the build is reproducible), and the per-kernel hashes of THAT library go into the file as
kernel_code_sha16.  usage: python profiles/scripts/r05_pmc_code_ids.py [commit]"""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
commit = sys.argv[1] if len(sys.argv) > 1 else "3e8085e"
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(path))
with tempfile.TemporaryDirectory() as tmp:
    tar = subprocess.run(["git", "-C", ROOT, "archive", commit], check=True, stdout=subprocess.PIPE).stdout
    subprocess.run(["tar", "-x", "-C", tmp], input=tar, check=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(tmp, "hal_amd", "csrc"), "lib"])
    lib = os.path.join(tmp, "hal_amd", "libhgx.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    if sha != d["libhgx_sha16"]:
        sys.exit("the rebuilt library of %s is %s, the PMC run measured %s: nothing written" % (commit, sha, d["libhgx_sha16"]))
    ids = bench.kernel_code_sha16s(lib)
measured = set(d.get("kernels", {})) | set(d.get("rotating", {}))
d["kernel_code_sha16"] = {k: v for k, v in sorted(ids.items()) if k in measured}
d["kernel_code_source"] = ("profiles/scripts/r05_pmc_code_ids.py: commit %s rebuilt, libhgx.so sha256 %s... = the measured library's; the hash of every "
                           "measured kernel's machine code in it (bench.kernel_code_sha16s)" % (commit, sha))
json.dump(d, open(path, "w"), indent=1, sort_keys=True)
now = bench.kernel_code_sha16s()
same = sorted(k for k in d["kernel_code_sha16"] if now.get(k) == d["kernel_code_sha16"][k])
print("measured kernels whose code is unchanged in the present library (%d of %d): %s" % (len(same), len(d["kernel_code_sha16"]), " ".join(same)))
print("changed since: %s" % " ".join(sorted(set(d["kernel_code_sha16"]) - set(same))))
