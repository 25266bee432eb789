"""Stress of hgx_liftover_submit / _collect: two plans, batches of changing sizes and interval lengths (buffer growth and the
repeat paths inside collect), each checked against hgx_liftover_run_device on a third plan."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hal_amd
from bench import workload_options

al = hal_amd.Alignment.random(workload_options(0.05, "cfg2"), device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
NMAX = 400000
ref = hal_amd.LiftoverPlan(al, src, tgt, max_queries=NMAX)
plans = [hal_amd.LiftoverPlan(al, src, tgt, max_queries=NMAX) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
g = torch.Generator().manual_seed(3)
t0, rounds, recs = time.time(), 0, 0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
pend = [None, None]


def check(k):
    global recs
    i, (gs, ge, st) = pend[k]
    ptr, n = plans[k].collect()
    with torch.cuda.stream(streams[k]):
        got = plans[k].records_to_tensor(ptr, n).cpu()
    p2, n2 = ref.run(gs, ge, st)
    want = ref.records_to_tensor(p2, n2).cpu()
    assert n == n2 and torch.equal(got, want), (i, n, n2)
    recs += n
    pend[k] = None


while time.time() - t0 < budget:
    n = int(torch.randint(1, NMAX, (1,), generator=g))
    if rounds % 7 == 3:
        n = int(torch.randint(1, 300, (1,), generator=g))
    maxlen = [300, 1000, 5000, 20000][int(torch.randint(0, 4, (1,), generator=g))]
    starts = torch.randint(0, length - maxlen - 1, (n,), generator=g)
    lens = torch.randint(1, maxlen, (n,), generator=g)
    st = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8).cuda()
    batch = ((starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), st)
    k = rounds & 1
    if pend[k] is not None:
        check(k)
    plans[k].submit(*batch, stream=streams[k])
    pend[k] = (rounds, batch)
    rounds += 1
for k in range(2):
    if pend[k] is not None:
        check(k)
print("in-flight stress: %d batches, %d records, all identical to hgx_liftover_run_device (%.0f s)" % (rounds, recs, time.time() - t0))
