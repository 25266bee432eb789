"""Kernel times of one cfg2 batch (no checks): for quick experiments on a modified library."""
import sys
sys.path.insert(0, '.')
import torch, hal_amd, bench
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
n = 1000000
starts, lens, strand = bench.make_queries(length, n, 1234)
gs = (starts + ss).cuda(); ge = (starts + lens - 1 + ss).cuda(); st = strand.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=n)
for _ in range(5):
    plan.run(gs, ge, st)
st = plan.stats()
print({k: round(v["ms"], 4) for k, v in plan.kernel_times().items()}, st["total_ms"], "table:", st["composed_kind"], st["composed_records"], round(st["composed_build_ms"], 2), "ms")
