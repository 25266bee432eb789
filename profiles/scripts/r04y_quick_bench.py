"""The last session's host-side changes on a GPU, in the few GPU seconds the round had left (no torch: its first import on a fresh box
costs more than the whole budget): config 3 end to end (hal2maf over the whole reference genome, twice), hal2maf --unique (2 M columns),
halAlignmentDepth's wig text end to end (twice), the text path on 1 M BED6 lines (best of 3) and 200 k BED12 lines to PSL.
usage: python profiles/scripts/r04y_quick_bench.py [scale]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import random
import numpy as np
pyrand = random.Random(11)
import hal_amd
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
out = {"scale": scale}


def leg(name):
    print(json.dumps({name: out[name]}), flush=True)

t0 = time.perf_counter()
opts = hal_amd.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50, max_segment_length=200,
                           min_segments=int(700000 * scale), max_segments=int(1400000 * scale), seed=2, with_dna="fast")
al = hal_amd.Alignment.random(opts, device=0)
out["generate_s"] = time.perf_counter() - t0
leg("generate_s")
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
seq_name, _, length = al.sequences(src)[0]
ncol = al.genome_length(src)
os.environ["HGX_MAF_TIMING"] = "1"
al.maf_export_bytes(src, start=0, length=200000, no_ancestors=True)
runs = []
for _ in range(2):
    t0 = time.perf_counter()
    nb = al.maf_export_bytes(src, no_ancestors=True)
    runs.append(time.perf_counter() - t0)
out["hal2maf_full"] = {"columns": ncol, "runs_seconds": runs, "columns_per_s": ncol / min(runs), "maf_bytes": nb}
leg("hal2maf_full")
del os.environ["HGX_MAF_TIMING"]
ncu = min(2000000, ncol)
al.maf_export_bytes(src, 0, start=0, length=200000, no_ancestors=True, unique=True)
t0 = time.perf_counter()
nb = al.maf_export_bytes(src, 0, start=0, length=ncu, no_ancestors=True, unique=True)
out["hal2maf_unique_2M"] = {"columns_per_s": ncu / (time.perf_counter() - t0), "maf_bytes": nb}
leg("hal2maf_unique_2M")
al.alignment_depth_bytes(src, length=min(ncol, 2000000))
runs = []
for _ in range(2):
    t0 = time.perf_counter()
    nb = al.alignment_depth_bytes(src)
    runs.append(time.perf_counter() - t0)
out["depth_wig"] = {"columns": ncol, "runs_seconds": runs, "columns_per_s": ncol / min(runs), "wig_bytes": nb}
leg("depth_wig")
rs = np.random.default_rng(1)
n = 1000000
lens = rs.integers(50, 1001, n)
starts = (rs.random(n) * (length - 1001)).astype(np.int64)
strand = rs.integers(0, 2, n)
bed = "".join("%s\t%d\t%d\tq%d\t0\t%s\n" % (seq_name, s, s + l, i, "+-"[k]) for i, (s, l, k) in enumerate(zip(starts.tolist(), lens.tolist(), strand.tolist()))).encode()
os.environ["HGX_TEXT_TIMING"] = "1"
hal_amd.liftover_convert_bytes(al, src, bed, tgt, count_lines=False)
runs = []
for _ in range(3):
    t0 = time.perf_counter()
    nb, _ = hal_amd.liftover_convert_bytes(al, src, bed, tgt, count_lines=False)
    runs.append(time.perf_counter() - t0)
del os.environ["HGX_TEXT_TIMING"]
out["end_to_end"] = {"lines": n, "runs_seconds": runs, "intervals_per_s": n / min(runs), "bytes_out": nb}
leg("end_to_end")
npsl = 200000
b12 = []
for i, (a0, l0) in enumerate(zip(starts[:npsl].tolist(), lens[:npsl].tolist())):
    cut = sorted(pyrand.sample(range(1, l0), 3))
    b12.append("%s\t%d\t%d\tn%d\t0\t%s\t%d\t%d\t0\t2\t%d,%d,\t0,%d," % (seq_name, a0, a0 + l0, i, "+-"[i & 1], a0, a0 + l0, cut[0], l0 - cut[1], cut[1]))
psl_in = ("\n".join(b12) + "\n").encode()
hal_amd.liftover_convert_bytes(al, src, ("\n".join(b12[:2000]) + "\n").encode(), tgt, out_psl=True)
t0 = time.perf_counter()
nb, nl = hal_amd.liftover_convert_bytes(al, src, psl_in, tgt, out_psl=True)
dt = time.perf_counter() - t0
out["liftover_psl"] = {"lines_in": npsl, "seconds": dt, "lines_per_s": npsl / dt, "bytes_out": nb, "lines_out": nl}
leg("liftover_psl")
