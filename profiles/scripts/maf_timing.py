"""hal2maf end to end on the GPU path (BASELINE config 3, scaled): time per stage."""
import sys, time
sys.path.insert(0, '.')
import hal_amd, bench
o = bench.workload_options(0.1)
o.with_dna = True
t = time.time(); al = hal_amd.Alignment.random(o, device=0); print("generate %.1fs" % (time.time() - t))
g = al.genome_id("Genome_9"); n = al.genome_length(g)
for kw in (dict(no_ancestors=True), dict()):
    t = time.time(); maf = al.maf_export(g, **kw); dt = time.time() - t
    print("hal2maf --refGenome Genome_9 %s: %d columns, %.2f s, %.2f M columns/s, %d blocks, %.1f MB" %
          (kw, n, dt, n / dt / 1e6, maf.count("\na\n") + 1, len(maf) / 1e6))
t = time.time(); off, rows = al.column_rows(g, 0, min(n, 2000000)); dt = time.time() - t
print("column_rows 2M columns: %.2fs, %d rows" % (dt, len(rows)))
