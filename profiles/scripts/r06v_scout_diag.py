"""k_lift_classify's scouts against the list form and the inline form: one plan batch after batch and two batches in flight, timed by the
wall clock and by the plan's own events.  usage: python r06v_scout_diag.py [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hal_amd, bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
al = hal_amd.Alignment.random(bench.workload_options(1.0), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
n = al.genome_length(src)
batches = []
for seed in range(4):
    s, l, st = bench.make_queries(n, 1000000, 1234 + seed)
    batches.append((s.cuda(), (s + l - 1).cuda(), st.cuda()))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def sync(): torch.cuda.synchronize()
for mode in sys.argv[2:] or ["scouts", "list", "inline"]:
    os.environ.pop("HGX_LIFT_SCOUT", None)
    if mode == "list":
        os.environ["HGX_LIFT_SCOUT"] = "0"
    plans = [hal_amd.LiftoverPlan(al, src, tgt, max_queries=1000000) for _ in range(2)]
    for p in plans:
        if mode == "inline":
            p.set_workers(0)
        for i in range(4):
            p.run(*batches[i])
        p.set_timing(0)
    sync()
    t = time.perf_counter()
    for i in range(steps):
        plans[0].run(*batches[i & 3])
    sync()
    one = (time.perf_counter() - t) / steps
    pending = [False, False]
    def two(k_steps):
        for i in range(k_steps):
            k = i & 1
            if pending[k]:
                plans[k].collect(); pending[k] = False
            plans[k].submit(*batches[i & 3], stream=streams[k]); pending[k] = True
        for k in (0, 1):
            if pending[k]:
                plans[k].collect(); pending[k] = False
    two(10); sync()
    t = time.perf_counter(); two(steps); sync()
    fl = (time.perf_counter() - t) / steps
    plans[0].set_timing(2)
    for i in range(steps):
        plans[0].run(*batches[i & 3])
    kt = plans[0].kernel_times()
    st = plans[0].stats()
    print(mode, "one plan %.4f ms  two in flight %.4f ms  events(one plan):" % (one * 1e3, fl * 1e3),
          {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in kt.items()}, "general", st["general_queries"], "records", st["records"], flush=True)
    del plans
