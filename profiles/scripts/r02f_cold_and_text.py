"""Where the cold pass and the text path spend their time: table build phases (HGX_BUILD_TIMING), plan creation in a process
that has already loaded its code objects, and the stages of hgx_liftover_convert (HGX_TEXT_TIMING)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hal_amd
from bench import workload_options, make_queries

os.environ["HGX_BUILD_TIMING"] = "1"
small = hal_amd.Alignment.random(workload_options(0.01, "cfg2"), device=0)
s9, s2 = small.genome_id("Genome_9"), small.genome_id("Genome_2")
_, ss, length = small.sequences(s9)[0]
st, ln, sd = make_queries(length, 20000, 1)
print("--- small alignment (loads the code objects)", flush=True)
p = hal_amd.LiftoverPlan(small, s9, s2, max_queries=20000)
p.run((st + ss).cuda(), (st + ln - 1 + ss).cuda(), sd.cuda())
del p
al = hal_amd.Alignment.random(workload_options(1.0, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
seq_name, ss, length = al.sequences(src)[0]
nq = 1000000
starts, lens, strand = make_queries(length, nq, 1234)
gs, ge, sdv = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
torch.cuda.synchronize()
print("--- cfg2, cold pass in a warm process", flush=True)
t0 = time.perf_counter()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
torch.cuda.synchronize()
t1 = time.perf_counter()
plan.run(gs, ge, sdv)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("plan create %.2f ms, first run (table build + lookup) %.2f ms, table %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, plan.stats()["composed_build_ms"]), flush=True)
del os.environ["HGX_BUILD_TIMING"]
os.environ["HGX_TEXT_TIMING"] = "1"
sn, lnn, tn = starts.numpy(), lens.numpy(), strand.numpy()
bed = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (seq_name, int(a), int(a + b), chr(int(c))) for a, b, c in zip(sn, lnn, tn)).encode()
for _ in range(4):
    t0 = time.perf_counter()
    nb, nl = hal_amd.liftover_convert_bytes(al, src, bed, tgt)
    print("convert: %.1f ms (incl. Python's line count), %d bytes" % ((time.perf_counter() - t0) * 1e3, nb), flush=True)
