"""Kernel times of one cfg2-shaped batch in the plan's modes (merged single-pass / unmerged table / walk), and a record-for-record
comparison between them.  Usage: python profiles/scripts/r02_plan_modes.py [scale] [queries] [workload]"""
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
al = hal_amd.Alignment.random(workload_options(scale, workload), device=0)
src, tgt = al.genome_id("Genome_9" if workload == "cfg2" else "Genome_44"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
starts, lens, strand = make_queries(length, nq, 1234)
gs, ge, st = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
ref = None
for name, env in (("merged", {"HGX_COMPOSED_UP": "1"}), ("through", {"HGX_COMPOSED_UP": "1", "HGX_MERGED": "0"}), ("walk", {"HGX_COMPOSED_UP": "0"})):
    for k in ("HGX_COMPOSED_UP", "HGX_MERGED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if name == "through":  # a fresh image: the table cache is per alignment
        al2 = hal_amd.Alignment.random(workload_options(scale, workload), device=0)
    else:
        al2 = al
    t0 = time.perf_counter()
    plan = hal_amd.LiftoverPlan(al2, src, tgt, max_queries=nq)
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    ptr, nrec = plan.run(gs, ge, st)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    recs = plan.records_to_tensor(ptr, nrec).clone()
    for _ in range(5):
        plan.run(gs, ge, st)
    plan.set_timing(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.run(gs, ge, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    kt = {k: round(v["ms"] / 20, 4) for k, v in plan.kernel_times().items()}
    stt = plan.stats()
    print(name, "create %.1f ms, first run %.1f ms, steady %.3f ms/step, %.1f M intervals/s, records %d, kind %d, table %d recs (%.1f ms), deferred %d"
          % (t_create * 1e3, t_first * 1e3, dt * 1e3, nq / dt / 1e6, nrec, stt["composed_kind"], stt["composed_records"], stt["composed_build_ms"],
             stt["deferred_queries"]), kt, flush=True)
    if ref is None:
        ref = recs
    else:
        print("   identical to merged:", bool(recs.shape == ref.shape and torch.equal(recs, ref)), flush=True)
    del plan
