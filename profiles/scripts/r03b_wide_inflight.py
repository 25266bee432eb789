#!/usr/bin/env python3
"""Two batches in flight on int64 tables (HGX_FORCE_WIDE=1) and on int32 tables, cfg2: step times, and — under
rocprofv3 --kernel-trace — the launches' start and end times, to see how much the two plans' launches overlap.
usage: r03b_wide_inflight.py <wide 0|1> [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
wide = sys.argv[1] == "1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
if wide:
    os.environ["HGX_FORCE_WIDE"] = "1"
import torch
import hal_amd
import bench

al = hal_amd.Alignment.random(bench.workload_options(1.0, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
nq = 1000000
starts, lens, strand = bench.make_queries(length, nq, 1234)
dev = torch.device("cuda", 0)
gs, ge, st = (starts + ss).to(dev), (starts + lens - 1 + ss).to(dev), strand.to(dev)
plans = [hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq) for _ in range(2)]
for p in plans:
    for _ in range(3):
        p.run(gs, ge, st)
    p.set_timing(0)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
pending = [False, False]


def two(k):
    for i in range(k):
        j = i & 1
        if pending[j]:
            plans[j].collect()
            pending[j] = False
        plans[j].submit(gs, ge, st, stream=streams[j])
        pending[j] = True
    for j in (0, 1):
        if pending[j]:
            plans[j].collect()
            pending[j] = False


two(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
two(steps)
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
for _ in range(5):
    plans[0].run(gs, ge, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    plans[0].run(gs, ge, st)
torch.cuda.synchronize()
dt1 = time.perf_counter() - t0
print("wide=%d two in flight %.4f ms/step, one plan %.4f ms/step, kind %d" % (wide, 1e3 * dt2 / steps, 1e3 * dt1 / steps, plans[0].stats()["composed_kind"]))
