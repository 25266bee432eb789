"""one failing soak case of hal2maf --global, recorded for the replay harness: multiseq seed 1431 (6 genomes, root 807 bases)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mode, path = sys.argv[1], sys.argv[2]
os.environ["HGX_MAF_DUMP" if mode == "dump" else "HGX_MAF_REPLAY"] = path
os.environ.setdefault("HGX_LIB_PATH", os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so"))
import halfix
import hal_amd as hal
img = "/tmp/case1431.hgx"
halfix.write_hgx(img, halfix.random_multiseq_alignment(1431, n_genomes=6, max_children=2, root_len=807))
al = hal.Alignment.open(img, device=0 if mode == "dump" else -1)
got = al.maf_export_global()
out = "/tmp/case1431.oracle.maf"
subprocess.check_call([os.path.join(ROOT, "oracle", "_build", "hal_oracle"), "maf", img, out, "--global"])
want = open(out).read()
open("/tmp/case1431.got.maf", "w").write(got)
print("same" if got == want else "DIFFERENT", len(got), len(want))
