"""N plans, N streams, a batch each in flight (hgx_liftover_submit / _collect) against one plan run batch after batch:
the cfg2 batch, steady state.  Usage: python profiles/scripts/r02t_pipelined.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hal_amd
from bench import workload_options, make_queries

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # plans = batches in flight
nq = 1000000
al = hal_amd.Alignment.random(workload_options(1.0, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
q = []
for seed in (1234, 99, 7, 8)[:NP]:
    starts, lens, strand = make_queries(length, nq, seed)
    q.append(((starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()))
plans = [hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq) for _ in range(NP)]
streams = [torch.cuda.Stream() for _ in range(NP)]
for p, a in zip(plans, q):
    for _ in range(12):
        p.run(*a)
    p.set_timing(0)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(steps):
        plans[0].run(*q[0])
    torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    pend = [False] * NP
    for i in range(steps):
        k = i % NP
        if pend[k]:
            plans[k].collect()
        plans[k].submit(*q[k], stream=streams[k])
        pend[k] = True
    for k in range(NP):
        if pend[k]:
            plans[k].collect()
    torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / steps
    print("one plan, batch after batch: %.4f ms per step = %.0f M intervals/s; %d in flight: %.4f ms = %.0f M intervals/s"
          % (t_sync * 1e3, nq / t_sync / 1e6, NP, t_pipe * 1e3, nq / t_pipe / 1e6), flush=True)
