"""one failing soak case, recorded for the replay harness: multiseq seed 412 (8 genomes, a chain), hal2maf --refGenome G5 --maxBlockLen 17"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mode, path = sys.argv[1], sys.argv[2]
os.environ["HGX_MAF_DUMP" if mode == "dump" else "HGX_MAF_REPLAY"] = path
os.environ.setdefault("HGX_LIB_PATH", os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so"))
import halfix
import hal_amd as hal
img = "/tmp/case412.hgx"
halfix.write_hgx(img, halfix.random_multiseq_alignment(412, n_genomes=8, max_children=1, root_len=1108))
al = hal.Alignment.open(img, device=0 if mode == "dump" else -1)
got = al.maf_export(al.genome_id("G5"), max_block_len=17)
out = "/tmp/case412.oracle.maf"
subprocess.check_call([os.path.join(ROOT, "oracle", "_build", "hal_oracle"), "maf", img, out, "--refGenome", "G5", "--maxBlockLen", "17"])
want = open(out).read()
open("/tmp/case412.got.maf", "w").write(got)
print("same" if got == want else "DIFFERENT", len(got), len(want))
