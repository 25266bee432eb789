"""End-to-end hal2maf / halAlignmentDepth CLI timings on a 1/10-scale config-2 alignment with DNA."""
import os, subprocess, tempfile, time
B = "hal_amd/_build"
with tempfile.TemporaryDirectory() as T:
    a = os.path.join(T, "a.hgx")
    subprocess.check_call([B + "/hgxRandGen", "--minGenomes", "2", "--maxGenomes", "10", "--meanDegree", "1.5", "--minSegmentLength", "50",
                           "--maxSegmentLength", "200", "--minSegments", "70000", "--maxSegments", "140000", "--maxBranchLength", "3",
                           "--seed", "2", a], stderr=subprocess.DEVNULL)
    for opts in (["--noAncestors"], []):
        t = time.time()
        subprocess.check_call([B + "/hal2maf", "--refGenome", "Genome_9"] + opts + [a, os.path.join(T, "o.maf")], env=dict(os.environ, HGX_MAF_TIMING="1"))
        dt = time.time() - t
        print("hal2maf --refGenome Genome_9 %s: %.2f s wall incl. image load/upload, %.1f MB, %.2f M columns/s" %
              (" ".join(opts), dt, os.path.getsize(os.path.join(T, "o.maf")) / 1e6, 5467200 / dt / 1e6))
    t = time.time()
    subprocess.check_call([B + "/halAlignmentDepth", a, "Genome_9", "--outWiggle", os.path.join(T, "o.wig")])
    dt = time.time() - t
    print("halAlignmentDepth Genome_9: %.2f s wall, %.1f M columns/s" % (dt, 5467200 / dt / 1e6))
