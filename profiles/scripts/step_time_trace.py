"""Wall time of consecutive liftover steps in a fresh process (is there a start-up transient on the host side?)."""
import sys, time
sys.path.insert(0, '.')
import torch, hal_amd, bench
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
n = 1000000
starts, lens, strand = bench.make_queries(length, n, 1234)
gs = (starts + ss).cuda(); ge = (starts + lens - 1 + ss).cuda(); st = strand.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=n)
ts = []
for i in range(60):
    t0 = time.perf_counter(); plan.run(gs, ge, st); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.2f" % t for t in ts))
print("device total_ms", plan.stats()["total_ms"])
