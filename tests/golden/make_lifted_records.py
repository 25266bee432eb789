"""Makes tests/golden/lifted_records.npz on a GPU box: the hgx_record rows of a real lift (halRandGen-shaped alignment, seed 2,
Genome_9 -> Genome_2, 600 BED6 intervals with '+', '-' and '.' strands) together with the device's wire blobs of two shards
of it — the data the world-size-2 gloo test exchanges.  Usage: python tests/golden/make_lifted_records.py <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hal_amd  # noqa: E402

opts = hal_amd.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10, max_segment_length=60,
                           min_segments=200, max_segments=600, seed=2, with_dna=False)
al = hal_amd.Alignment.random(opts, device=0)
src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
_, ss, length = al.sequences(src)[0]
n = 600
g = torch.Generator().manual_seed(3)
starts = torch.randint(0, length - 400, (n,), generator=g)
lens = torch.randint(1, 400, (n,), generator=g)
strand = torch.tensor([ord("+-."[i % 3]) for i in range(n)], dtype=torch.uint8)
gs, ge, st = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
plan = hal_amd.LiftoverPlan(al, src, tgt, max_queries=n)
ptr, nrec = plan.run(gs, ge, st)
whole = plan.records_to_tensor(ptr, nrec).cpu().numpy()
blobs = []
for lo, hi in ((0, 250), (250, 600)):  # two uneven shards, as two ranks would lift them
    ptr, k = plan.run(gs[lo:hi].contiguous(), ge[lo:hi].contiguous(), st[lo:hi].contiguous())
    shard = plan.records_to_tensor(ptr, k).cpu().numpy()
    blob, fmt = plan.wire_blob(first_query=lo)
    blobs.append((shard, blob.cpu().numpy(), fmt))
np.savez_compressed(sys.argv[1], whole=whole, shard0=blobs[0][0], blob0=blobs[0][1], shard1=blobs[1][0], blob1=blobs[1][1],
                    formats=np.array([blobs[0][2], blobs[1][2]]), bounds=np.array([0, 250, 600]))
print("records", whole.shape, "blob bytes", blobs[0][1].shape, blobs[1][1].shape, "formats", blobs[0][2], blobs[1][2])
