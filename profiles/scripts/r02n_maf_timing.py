import sys, time, os
sys.path.insert(0, '.')
import hal_amd, bench
o = bench.workload_options(1.0, "cfg2", dna=True)
al = hal_amd.Alignment.random(o, device=0)
g = al.genome_id("Genome_9")
al.maf_export_bytes(g, start=0, length=200000, no_ancestors=True)
os.environ["HGX_MAF_TIMING"] = "1"
for rep in range(2):
    t = time.time(); m = al.maf_export_bytes(g, start=0, length=8000000, no_ancestors=True); dt = time.time() - t
    print("8M columns: %.3f s, %.1f M columns/s" % (dt, 8 / dt), flush=True)
