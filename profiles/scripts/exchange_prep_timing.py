"""What a rank does per step besides mapping when N > 1 (measurable on one GPU): copy the records out of the plan and
pack them to the 20-byte wire form."""
import sys, time
sys.path.insert(0, '.')
import torch, hal_amd, bench
from hal_amd import shard
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
n = 1000000
starts, lens, strand = bench.make_queries(length, n, 1234)
gs = (starts + ss).cuda(); ge = (starts + lens - 1 + ss).cuda(); st = strand.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=n)
for _ in range(5):
    ptr, nrec = plan.run(gs, ge, st)
def t(f, k=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3, r
ms_copy, recs = t(lambda: plan.records_to_tensor(ptr, nrec))
ms_pack, packed = t(lambda: shard.pack_records(recs), 3)
ms_kpack, kp = t(lambda: plan.records_to_tensor(ptr, nrec, packed=True))
print("records %d: records_to_tensor %.3f ms, torch pack_records %.3f ms, packed copy by the library %.3f ms (%d -> %d MB)" % (nrec, ms_copy, ms_pack, ms_kpack, recs.numel() >> 20, packed.numel() >> 20))
