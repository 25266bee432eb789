"""Differential soak of k_lift_classify's worker workgroups: random halRandGen alignments, batches of tens of thousands of
intervals of mixed lengths, the records of a plan without workers against the records with few / many workers (and, on a
sample, against the oracle).  HGX_FORCE_WIDE=1 in the environment runs it on int64 tables.
usage: python profiles/scripts/r03_workers_soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hal_amd as hal

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "7")))
os.environ["HGX_COMPOSED_UP"] = "1"  # (the table from the first batch on)
t0 = time.time()
rounds = batches = generals = records = 0
while time.time() - t0 < budget:
    rounds += 1
    seed = int(rng.integers(0, 10000))
    lo = int(rng.integers(5, 60)); hi = lo + int(rng.integers(1, 150))
    nlo = int(rng.integers(500, 4000)); nhi = nlo + int(rng.integers(1, 4000))
    opts = hal.RandOptions(mean_degree=float(rng.uniform(1.1, 2.5)), max_branch_length=float(rng.choice([1.5, 3.0, 5.0])), min_genomes=2,
                           max_genomes=int(rng.integers(3, 20)), min_segment_length=lo, max_segment_length=hi, min_segments=nlo, max_segments=nhi,
                           seed=seed, with_dna=False)
    try:
        al = hal.Alignment.random(opts, device=0)
    except hal.HgxError as e:
        if "runaway tree" in str(e):
            continue
        raise
    ng = al.num_genomes
    for _ in range(3):
        src, tgt = int(rng.integers(0, ng)), int(rng.integers(0, ng))
        seqs = [q for q in al.sequences(src) if q[2] > 100]
        if not seqs:
            continue
        _, ss, length = seqs[int(rng.integers(0, len(seqs)))]
        n = int(rng.integers(20000, 70000))
        maxlen = int(rng.choice([40, 300, 3000]))
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        starts = torch.randint(0, max(1, length - 1), (n,), generator=g)
        lens = torch.randint(1, max(2, min(maxlen, length)), (n,), generator=g)
        gs, ge = (starts + ss).cuda(), (starts + lens - 1 + ss).clamp(max=ss + length - 1).cuda()
        st = torch.from_numpy(rng.choice(np.frombuffer(b"+-.", dtype=np.uint8), n)).cuda()
        nd = bool(rng.integers(0, 3) == 0)
        got = {}
        for workers in ("0", "5", "200"):
            os.environ["HGX_LIFT_WORKERS"] = workers
            plan = hal.LiftoverPlan(al, src, tgt, max_queries=n, traverse_dupes=not nd)
            try:
                ptr, nrec = plan.run(gs, ge, st)
            except hal.HgxError as e:  # (a batch that expands to more records than the device holds)
                if "out of memory" in str(e) or "2^32" in str(e):
                    break
                raise
            got[workers] = plan.records_to_tensor(ptr, nrec).cpu()
            if workers == "0":
                generals += plan.stats()["general_queries"]
                records += nrec
                if plan.stats()["composed_kind"] != 3:
                    break
        else:
            batches += 1
            assert torch.equal(got["0"], got["5"]) and torch.equal(got["0"], got["200"]), (seed, src, tgt, n, maxlen, nd)
print("workers soak%s: %d alignments, %d batches of 20-90 k intervals, %d general intervals, %d records: the same with 0, 5 and 200 workers (%d s)"
      % (" (int64 tables)" if os.environ.get("HGX_FORCE_WIDE") else "", rounds, batches, generals, records, time.time() - t0))
