"""Workload of the PMC passes (profiles/scripts/r05_pmc.py): the bench's cfg2 batch through the default plan (merged table,
single-pass kernels), through the level walk, and the depth kernel over the source genome — a few launches of every kernel.
--form rotating: bench.py's timed loop (`value`) instead — four distinct batches, a plan each, taken in turn (only the single-pass
kernels; no walk, no columns, no calibration launches)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hal_amd
from bench import workload_options, make_queries

pos = [a for a in sys.argv[1:] if not a.startswith("--") and a not in ("steady", "rotating")]
scale = float(pos[0]) if len(pos) > 0 else 1.0
nq = int(pos[1]) if len(pos) > 1 else 1000000
rotating = "--form" in sys.argv and sys.argv[sys.argv.index("--form") + 1] == "rotating"
al = hal_amd.Alignment.random(workload_options(scale, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
if rotating:
    K = 4
    plans, batches = [], []
    for k in range(K):
        s_k, l_k, d_k = make_queries(length, nq, 5000 + 17 * k)
        batches.append(((s_k + ss).cuda(), (s_k + l_k - 1 + ss).cuda(), d_k.cuda()))
        p = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
        p.set_workers(0)
        for _ in range(3):
            p.run(*batches[k])
        plans.append(p)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    pending = [False] * K
    for i in range(6 * K):
        k = i % K
        for j in ((i - 2) % K, k):
            if pending[j]:
                plans[j].collect()
                pending[j] = False
        plans[k].submit(*batches[k], stream=streams[i & 1])
        pending[k] = True
    for k in range(K):
        if pending[k]:
            plans[k].collect()
    torch.cuda.synchronize()
    sys.exit(0)
starts, lens, strand = make_queries(length, nq, 1234)
gs, ge, st = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
plan.set_workers(0)  # (the form the bench's timed region launches: general intervals finished by the wavefronts that meet them)
for _ in range(4):
    plan.run(gs, ge, st)
os.environ["HGX_COMPOSED_UP"] = "0"
walk = hal_amd.LiftoverPlan(al, src, tgt, max_queries=nq)
del os.environ["HGX_COMPOSED_UP"]
for _ in range(3):
    walk.run(gs, ge, st)
if "--no-columns" not in sys.argv:
    ncol = al.genome_length(src)
    out = torch.empty(ncol, dtype=torch.int32, device="cuda")
    for _ in range(2):
        al.columns_depth_device(src, 0, ncol, out.data_ptr())
torch.cuda.synchronize()
# calibration launches of known traffic, in the same passes (profiles/scripts/r03_pmc.py reads them back by kernel name):
#  * a 1 GiB device-to-device copy: 1 GiB read, 1 GiB written, streaming
#  * 16 Mi random 16-byte rows gathered from a 2 GiB table: one 128-byte line per row
if "--no-calibration" not in sys.argv:
    a = torch.empty(1 << 28, dtype=torch.int32, device="cuda").fill_(1)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    table = torch.empty((1 << 27, 4), dtype=torch.int32, device="cuda").fill_(2)
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    idx = torch.randint(0, 1 << 27, (1 << 24,), device="cuda", generator=g)
    for _ in range(3):
        rows = torch.index_select(table, 0, idx)
    torch.cuda.synchronize()
