"""phase profile of finish_wave (make profile-lib PROFILE_KERNEL=3; HGX_LIB_PATH) on the cfg2 batch, inline form"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hal_amd
from bench import workload_options, make_queries
al = hal_amd.Alignment.random(workload_options(1.0, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
nq = 1000000
st, ln, sd = make_queries(length, nq, 1234)
gs, ge, sdd = (st + ss).cuda(), (st + ln - 1 + ss).cuda(), sd.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
plan.set_workers(0)
for _ in range(6):
    plan.run(gs, ge, sdd)
print(plan.stats()["general_queries"])
