"""Where config 3's export spends its time on the GPU box: the profiling build (make hostprof-lib HOSTPROF_TICKS=1, HGX_LIB_PATH) with the
walk's tick counters, the export as it is, without rendering (HGX_MAF_NO_RENDER), with fewer rendering / describing threads and with
one batch ahead only; then the text path three times (the kept output blocks).  No torch.  usage: HGX_LIB_PATH=hal_amd/libhgx_hostprof.so
python profiles/scripts/r04y_maf_diag.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import hal_amd
opts = hal_amd.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50, max_segment_length=200,
                           min_segments=700000, max_segments=1400000, seed=2, with_dna="fast")
al = hal_amd.Alignment.random(opts, device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
seq_name, _, length = al.sequences(src)[0]
os.environ["HGX_MAF_TIMING"] = "1"
al.maf_export_bytes(src, start=0, length=4000000, no_ancestors=True)
VARIANTS = {"first": (("as it is", {}), ("as it is, again", {}), ("no rendering", {"HGX_MAF_NO_RENDER": "1"}), ("8 rendering threads", {"HGX_MAF_RENDER_THREADS": "8"}),
                     ("64 rendering threads", {"HGX_MAF_RENDER_THREADS": "64"}), ("2 describing threads", {"HGX_MAF_DESCRIBE_THREADS": "2"}),
                     ("one batch ahead", {"HGX_MAF_AHEAD": "1"})),
            "pool": (("as it is", {}), ("fresh logs", {"HGX_MAF_BATCH_POOL": "0"}), ("fresh logs, again", {"HGX_MAF_BATCH_POOL": "0"}), ("as it is, again", {}),
                     ("fresh logs, 16 rendering threads", {"HGX_MAF_BATCH_POOL": "0", "HGX_MAF_RENDER_THREADS": "16"}))}
VARIANTS["threads"] = tuple(("%d rendering threads" % n, {"HGX_MAF_RENDER_THREADS": str(n)}) for n in (32, 48, 64, 24, 32))
which = sys.argv[1] if len(sys.argv) > 1 else "first"
for label, env in VARIANTS[which]:
    os.environ.update(env)
    print("==", label, flush=True)
    sys.stderr.flush()
    t0 = time.perf_counter()
    nb = al.maf_export_bytes(src, no_ancestors=True)
    print("   export %.3f s, %d bytes" % (time.perf_counter() - t0, nb), flush=True)
    for k in env:
        del os.environ[k]
del os.environ["HGX_MAF_TIMING"]
if which != "first":
    sys.exit(0)
rs = np.random.default_rng(1)
n = 1000000
lens = rs.integers(50, 1001, n)
starts = (rs.random(n) * (length - 1001)).astype(np.int64)
strand = rs.integers(0, 2, n)
bed = "".join("%s\t%d\t%d\tq%d\t0\t%s\n" % (seq_name, s, s + l, i, "+-"[k]) for i, (s, l, k) in enumerate(zip(starts.tolist(), lens.tolist(), strand.tolist()))).encode()
os.environ["HGX_TEXT_TIMING"] = "1"
for _ in range(4):
    t0 = time.perf_counter()
    nb, _ = hal_amd.liftover_convert_bytes(al, src, bed, tgt, count_lines=False)
    print("   text path %.2f ms, %d bytes" % (1e3 * (time.perf_counter() - t0), nb), flush=True)
