"""(1) PCIe-inclusive rates of the host-buffer entry points on cfg2; (2) BASELINE config 4 shape on one GPU:
50-genome alignment, Genome_44 -> Genome_2, 1.25 M intervals (one rank's shard of the 10 M)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, hal_amd, bench
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
name, ss, length = al.sequences(src)[0]
n = 1000000
starts, lens, strand = bench.make_queries(length, n, 1234)
iv = np.zeros(n, dtype=hal_amd.api.INTERVAL_DTYPE)
iv["start"] = starts.numpy(); iv["end"] = (starts + lens).numpy(); iv["strand"] = np.where(strand.numpy() == ord("+"), b"+", b"-")
for _ in range(2):
    t = time.time(); recs = al.liftover_batch(src, tgt, iv); dt = time.time() - t
print("hgx_liftover_batch (host buffers in/out, plan + H2D + run + D2H): %d intervals, %.3f s, %.1f M intervals/s, %d records" % (n, dt, n / dt / 1e6, len(recs)))
bed = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (name, int(s), int(s + l), chr(int(c))) for s, l, c in zip(starts[:n], lens[:n], strand[:n]))
for _ in range(2):
    t = time.time(); out = hal_amd.liftover_convert(al, src, bed, tgt); dt = time.time() - t
print("hgx_liftover_convert (BED text in/out): %d lines in, %d lines out, %.3f s, %.2f M intervals/s" % (n, out.count("\n"), dt, n / dt / 1e6))
del al
t = time.time()
o = hal_amd.RandOptions(mean_degree=2.0, max_branch_length=3.0, min_genomes=2, max_genomes=50, min_segment_length=50, max_segment_length=200, min_segments=700000, max_segments=1400000, seed=0, with_dna=False)
al = hal_amd.Alignment.random(o, device=0)
print("cfg4 alignment: %d genomes, generated in %.1f s" % (al.num_genomes, time.time() - t))
src, tgt = al.genome_id("Genome_44"), al.genome_id("Genome_2")
name, ss, length = al.sequences(src)[0]
n = 1250000
starts, lens, strand = bench.make_queries(length, n, 99)
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=n)
gs, ge, st = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
def report(tag):
    s = plan.stats()
    print("cfg4 shard on 1 GPU, %s: %d intervals Genome_44->Genome_2 (7 up, 1 down): %.3f ms, %.1f M intervals/s, %d records, %d mapped pieces, table kind %d (%d records, built in %.1f ms)"
          % (tag, n, s["total_ms"], n / s["total_ms"] / 1e3, nrec, s["mapped_pieces"], s["composed_kind"], s["composed_records"], s["composed_build_ms"]),
          {k: round(v["ms"], 3) for k, v in plan.kernel_times().items()})
for _ in range(3):
    ptr, nrec = plan.run(gs, ge, st)
report("level walk (first runs of a plan)")
for _ in range(12):  # the plan changes over to its table after 4 intervals per source segment
    ptr, nrec = plan.run(gs, ge, st)
report("after the change-over")
