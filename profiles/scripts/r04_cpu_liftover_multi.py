"""hgx_liftover_convert_multi on a machine without a GPU: the same alignment handle n times over (device -1; the profiling build takes
every group's records from HGX_LIFT_REPLAY), so that the sharing out of the input's lines over several handles — shares, rounds, the
collation of the shares' texts — runs against the oracle's text.  One conversion a process.
usage: HGX_LIB_PATH=hal_amd/libhgx_hostprof.so HGX_LIFT_REPLAY=rec.bin python r04_cpu_liftover_multi.py img src in.bed tgt out.bed n [--noDupes] [--bedType N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hal_amd
img, src, bed, tgt, out, n = sys.argv[1:7]
opts = sys.argv[7:]
al = hal_amd.Alignment.open(img, device=-1)
kw = dict(traverse_dupes="--noDupes" not in opts)
if "--bedType" in opts:
    kw["bed_type"] = int(opts[opts.index("--bedType") + 1])
try:
    text = hal_amd.liftover_convert_multi([al] * int(n), al.genome_id(src), open(bed, "rb").read(), al.genome_id(tgt), **kw)
    rc = 0
except hal_amd.HgxError as e:
    text = e.partial_output
    sys.stderr.write("hal exception caught: %s\n" % e)
    rc = 1
open(out, "w").write(text)
sys.exit(rc)
