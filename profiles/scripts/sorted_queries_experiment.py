"""Experiment: does position-sorted query order speed the walk up? (same intervals, two orders)"""
import sys
sys.path.insert(0, '.')
import torch, hal_amd, bench
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
n = 1000000
starts, lens, strand = bench.make_queries(length, n, 1234)
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=n)
for name, order in (("random", torch.arange(n)), ("sorted", torch.argsort(starts))):
    gs = (starts[order] + ss).cuda(); ge = (starts[order] + lens[order] - 1 + ss).cuda(); st = strand[order].cuda()
    for _ in range(3):
        ptr, nrec = plan.run(gs, ge, st)
    s = plan.stats(); kt = plan.kernel_times()
    print(name, "total_ms %.3f walk_ms %.3f records %d" % (s["total_ms"], s["walk_ms"], nrec), {k: round(v["ms"], 3) for k, v in kt.items()})
