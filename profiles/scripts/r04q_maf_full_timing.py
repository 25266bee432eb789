"""config 3 (hal2maf --refGenome Genome_9 --noAncestors over all 54.7 M columns) through the library with its own timing lines"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hal_amd
from bench import workload_options
al = hal_amd.Alignment.random(workload_options(1.0, "cfg2", dna="fast"), device=0)
g = al.genome_id("Genome_9")
al.maf_export_bytes(g, start=0, length=200000, no_ancestors=True)
os.environ["HGX_MAF_TIMING"] = "1"
for _ in range(2):
    t0 = time.perf_counter()
    n = al.maf_export_bytes(g, no_ancestors=True)
    print("export %.3f s, %d bytes" % (time.perf_counter() - t0, n), flush=True)
