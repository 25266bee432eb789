"""cfg4 shard: how many pieces the intervals have that the counting launch passes on to k_locate_through + k_finish_lds / k_finish_big"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hal_amd
from bench import workload_options, make_queries
al = hal_amd.Alignment.random(workload_options(1.0, "cfg4"), device=0)
src, tgt = al.genome_id("Genome_44"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
nq = 1250000
st, ln, sd = make_queries(length, nq, 1234)
gs, ge, sdd = (st + ss).cuda(), (st + ln - 1 + ss).cuda(), sd.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
for _ in range(4):
    plan.run(gs, ge, sdd)
os.environ["HGX_LATE_HISTOGRAM"] = "1"
plan.run(gs, ge, sdd)
del os.environ["HGX_LATE_HISTOGRAM"]
print(plan.stats())
plan.set_timing(2)
for _ in range(5):
    plan.run(gs, ge, sdd)
print({k: round(v["ms"] / 5, 4) for k, v in plan.kernel_times().items()})
