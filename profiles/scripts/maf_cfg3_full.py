"""BASELINE config 3 at full size: hal2maf --refGenome Genome_9 --noAncestors over the whole deepest leaf of the config-2
alignment (with DNA), and halAlignmentDepth of the same genome, command-line tools, wall time."""
import os, subprocess, tempfile, time
B = "hal_amd/_build"
with tempfile.TemporaryDirectory() as T:
    a = os.path.join(T, "a.hgx")
    t = time.time()
    subprocess.check_call([B + "/hgxRandGen", "--minGenomes", "2", "--maxGenomes", "10", "--meanDegree", "1.5", "--minSegmentLength", "50",
                           "--maxSegmentLength", "200", "--minSegments", "700000", "--maxSegments", "1400000", "--maxBranchLength", "3",
                           "--seed", "2", a], stderr=subprocess.DEVNULL)
    print("alignment with DNA generated and written in %.1f s, %.2f GB" % (time.time() - t, os.path.getsize(a) / 1e9), flush=True)
    t = time.time()
    subprocess.check_call([B + "/hal2maf", "--refGenome", "Genome_9", "--noAncestors", a, os.path.join(T, "o.maf")],
                          env=dict(os.environ, HGX_MAF_TIMING="1"))
    dt = time.time() - t
    ncol = sum(1 for _ in [0])  # placeholder, real count from the timing line above
    print("hal2maf --refGenome Genome_9 --noAncestors: %.2f s wall incl. image load/upload, %.2f GB of MAF" %
          (dt, os.path.getsize(os.path.join(T, "o.maf")) / 1e9), flush=True)
    os.remove(os.path.join(T, "o.maf"))
    t = time.time()
    subprocess.check_call([B + "/halAlignmentDepth", a, "Genome_9", "--outWiggle", os.path.join(T, "o.wig")])
    dt = time.time() - t
    print("halAlignmentDepth Genome_9: %.2f s wall, %.2f GB of wiggle" % (dt, os.path.getsize(os.path.join(T, "o.wig")) / 1e9))
