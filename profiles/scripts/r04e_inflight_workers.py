import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hal_amd
from bench import workload_options, make_queries
al = hal_amd.Alignment.random(workload_options(1.0, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
nq = 1000000
st, ln, sd = make_queries(length, nq, 1234)
gs, ge, sdd = (st + ss).cuda(), (st + ln - 1 + ss).cuda(), sd.cuda()
plans = [hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq) for _ in range(2)]
for p in plans:
    for _ in range(4):
        p.run(gs, ge, sdd)
    p.set_timing(0)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def loop(n):
    pend = [False, False]
    for i in range(n):
        k = i & 1
        if pend[k]:
            plans[k].collect(); pend[k] = False
        plans[k].submit(gs, ge, sdd, stream=streams[k]); pend[k] = True
    for k in (0, 1):
        if pend[k]: plans[k].collect()
for rep in range(3):
    for w in (-1, 0, 150, 300, 600):
        for p in plans: p.set_workers(w)
        loop(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("workers %4d: %.4f ms/step  %.2f G/s" % (w, 1e3 * dt / 400, nq * 400 / dt / 1e9))
