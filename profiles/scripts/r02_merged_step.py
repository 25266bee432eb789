"""One cfg2-shaped batch on the default (merged single-pass) plan: step time without kernel events, then the kernel times.
Environment knobs (HGX_*) are taken as they are.  Usage: python profiles/scripts/r02_merged_step.py [scale] [queries] [workload] [tag]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hal_amd
from bench import workload_options, make_queries

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
tag = sys.argv[4] if len(sys.argv) > 4 else ""
os.environ.setdefault("HGX_COMPOSED_UP", "1")
al = hal_amd.Alignment.random(workload_options(scale, workload), device=0)
src, tgt = al.genome_id("Genome_9" if workload == "cfg2" else "Genome_44"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
starts, lens, strand = make_queries(length, nq, 1234)
gs, ge, st = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
t0 = time.perf_counter()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
torch.cuda.synchronize()
t_create = time.perf_counter() - t0
for _ in range(10):
    plan.run(gs, ge, st)
plan.set_timing(0)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.run(gs, ge, st)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
plan.set_timing(2)
for _ in range(20):
    ptr, nrec = plan.run(gs, ge, st)
kt = {k: round(v["ms"] / 20, 4) for k, v in plan.kernel_times().items()}
s = plan.stats()
env = {k: v for k, v in os.environ.items() if k.startswith("HGX_")}
print("%s %s: create %.1f ms (table %.1f ms, %d recs, %d flagged), step %.4f ms = %.0f M intervals/s, records %d, general %d, deferred %d, kernels %.4f ms %s"
      % (tag, env, t_create * 1e3, s["composed_build_ms"], s["composed_records"], s["composed_flagged"], best * 1e3, nq / best / 1e6, nrec,
         s["general_queries"], s["deferred_queries"], sum(kt.values()), kt), flush=True)
