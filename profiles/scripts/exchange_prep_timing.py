"""Host-side cost of the exchange step on one GPU (one-rank RCCL group): the wire blob, the count exchange, the payload
all-gather, against the bare run."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist, hal_amd, bench
from hal_amd import shard
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
n = 1000000
starts, lens, strand = bench.make_queries(length, n, 1234)
gs = (starts + ss).cuda(); ge = (starts + lens - 1 + ss).cuda(); st = strand.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=n)
for _ in range(12):
    ptr, nrec = plan.run(gs, ge, st)
col = shard.RecordCollator()
def timed(f, reps=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
print("run                     %.3f ms" % timed(lambda: plan.run(gs, ge, st)))
print("wire_blob               %.3f ms" % timed(lambda: plan.wire_blob(first_query=0)))
blob, fmt = plan.wire_blob(first_query=0)
print("all_gather_counts       %.3f ms" % timed(lambda: shard.all_gather_counts(blob.numel(), blob.device)))
def ex():
    col.wait(trim=False); col.submit(blob)
print("collator wait+submit    %.3f ms  (%d bytes, format %d)" % (timed(ex), blob.numel(), fmt))
def step():
    plan.run(gs, ge, st); b, _ = plan.wire_blob(first_query=0); col.wait(trim=False); col.submit(b)
print("whole step              %.3f ms" % timed(step))
col.drain()
dist.destroy_process_group()
