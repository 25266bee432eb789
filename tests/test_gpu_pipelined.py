"""hgx_liftover_submit / hgx_liftover_collect: two plans of one alignment with a batch each in flight on two streams give the
records of hgx_liftover_run_device, batch after batch — also for the batches that submit or collect run to the end themselves
(the first ones of a plan, which walk and then build the table; one with kernel events on)."""
import pytest

pytestmark = pytest.mark.gpu


def test_two_plans_two_streams(hal):
    import torch
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=20,
                           max_segment_length=80, min_segments=3000, max_segments=6000, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_8")
    _, ss, length = al.sequences(src)[0]
    n = 20000
    batches = []
    for b in range(6):
        g = torch.Generator().manual_seed(100 + b)
        starts = torch.randint(0, length - 400, (n,), generator=g)
        lens = torch.randint(1, 300, (n,), generator=g)
        st = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8).cuda()
        batches.append(((starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), st))
    ref_plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    want = []
    for gs, ge, st in batches:
        ptr, nrec = ref_plan.run(gs, ge, st)
        want.append(ref_plan.records_to_tensor(ptr, nrec).cpu())
    plans = [hal.LiftoverPlan(al, src, tgt, max_queries=n) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    got = [None] * len(batches)
    for rnd in range(3):  # (the same batches three times: the plans are in their steady state from the second round on)
        inflight = [None, None]
        for i, (gs, ge, st) in enumerate(batches):
            k = i & 1
            if inflight[k] is not None:
                ptr, nrec = plans[k].collect()
                with torch.cuda.stream(streams[k]):
                    got[inflight[k]] = plans[k].records_to_tensor(ptr, nrec).cpu()
            if rnd == 1 and i == 3:
                plans[k].set_timing(2)  # a batch that submit runs to the end itself
            plans[k].submit(gs, ge, st, stream=streams[k])
            plans[k].set_timing(0)
            inflight[k] = i
        for k in range(2):
            ptr, nrec = plans[k].collect()
            with torch.cuda.stream(streams[k]):
                got[inflight[k]] = plans[k].records_to_tensor(ptr, nrec).cpu()
        for i in range(len(batches)):
            assert torch.equal(got[i], want[i]), (rnd, i)
    assert plans[0].stats()["composed_kind"] == 3
    with pytest.raises(hal.HgxError):
        plans[0].collect()


def test_batches_with_long_intervals_in_flight(hal):
    """A 50-genome alignment whose intervals reach sets of more than 64 pieces (the LDS and global-scratch finishing kernels
    run inside the single-pass launch sequence, and collect repeats a batch whose scratch was too small): submit / collect
    against run_device."""
    import torch
    opts = hal.RandOptions(mean_degree=2.0, max_branch_length=3.0, min_genomes=2, max_genomes=50, min_segment_length=20,
                           max_segment_length=80, min_segments=1500, max_segments=3000, seed=0, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    names = [al.genome_name(g) for g in range(al.num_genomes)]
    src = al.genome_id("Genome_44") if "Genome_44" in names else al.num_genomes - 1
    tgt = al.genome_id("Genome_2")
    _, ss, length = al.sequences(src)[0]
    n = 30000
    batches = []
    for b in range(4):
        g = torch.Generator().manual_seed(200 + b)
        maxlen = (300, 3000, 300, 12000)[b]
        starts = torch.randint(0, max(1, length - maxlen - 1), (n,), generator=g)
        lens = torch.randint(1, maxlen, (n,), generator=g)
        st = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8).cuda()
        batches.append(((starts + ss).cuda(), (starts + lens - 1 + ss).clamp(max=ss + length - 1).cuda(), st))
    ref_plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    want = []
    for gs, ge, st in batches * 2:
        ptr, nrec = ref_plan.run(gs, ge, st)
        want.append(ref_plan.records_to_tensor(ptr, nrec).cpu())
    assert ref_plan.stats()["composed_kind"] == 3
    plans = [hal.LiftoverPlan(al, src, tgt, max_queries=n) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    inflight = [None, None]
    deferred = 0
    for i, (gs, ge, st) in enumerate(batches * 2):
        k = i & 1
        if inflight[k] is not None:
            ptr, nrec = plans[k].collect()
            deferred += plans[k].stats()["deferred_queries"]
            with torch.cuda.stream(streams[k]):
                assert torch.equal(plans[k].records_to_tensor(ptr, nrec).cpu(), want[inflight[k]]), inflight[k]
        plans[k].submit(gs, ge, st, stream=streams[k])
        inflight[k] = i
    for k in range(2):
        ptr, nrec = plans[k].collect()
        with torch.cuda.stream(streams[k]):
            assert torch.equal(plans[k].records_to_tensor(ptr, nrec).cpu(), want[inflight[k]]), inflight[k]
    assert ref_plan.stats()["general_queries"] > 0


def test_general_intervals_by_worker_workgroups(hal, monkeypatch):
    """k_lift_classify's workers (hgx_lift_kernels.hpp): the workgroups in front of the grid find the general intervals of the whole
    batch from the bucket entries and finish them while the tiles are classified.  Same records whether the general intervals
    are finished by the wavefronts that meet them (no workers: a plan's first run), by few or many workers, or — what a plan
    does by itself — by as many as its last run's count suggests."""
    import torch
    opts = hal.RandOptions(mean_degree=2.0, max_branch_length=3.0, min_genomes=2, max_genomes=50, min_segment_length=20,
                           max_segment_length=80, min_segments=1500, max_segments=3000, seed=0, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    names = [al.genome_name(g) for g in range(al.num_genomes)]
    src = al.genome_id("Genome_44") if "Genome_44" in names else al.num_genomes - 1
    tgt = al.genome_id("Genome_2")
    _, ss, length = al.sequences(src)[0]
    n = 70000
    batches = []
    for b, maxlen in enumerate((300, 3000, 60, 12000)):
        g = torch.Generator().manual_seed(300 + b)
        starts = torch.randint(0, max(1, length - maxlen - 1), (n,), generator=g)
        lens = torch.randint(1, maxlen, (n,), generator=g)
        st = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8).cuda()
        batches.append(((starts + ss).cuda(), (starts + lens - 1 + ss).clamp(max=ss + length - 1).cuda(), st))

    def run(plan):
        out = []
        for gs, ge, st in batches:
            ptr, nrec = plan.run(gs, ge, st)
            out.append(plan.records_to_tensor(ptr, nrec).cpu())
        return out

    monkeypatch.setenv("HGX_LIFT_WORKERS", "0")
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    want = run(plan)
    assert plan.stats()["composed_kind"] == 3 and plan.stats()["general_queries"] > 0
    # (round 6: the workers find their intervals themselves — "scouts", the default; HGX_LIFT_SCOUT=0: round 3's form, a pass over the
    # batch in front that lists them; HGX_LIFT_SCOUT_MAX: few scouts, whose shares take several rounds)
    for scout, most in (("1", None), ("0", None), ("1", "5")):
        monkeypatch.setenv("HGX_LIFT_SCOUT", scout)
        if most:
            monkeypatch.setenv("HGX_LIFT_SCOUT_MAX", most)
        for workers in ("3", "64", "1024"):
            monkeypatch.setenv("HGX_LIFT_WORKERS", workers)
            plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
            got = run(plan)
            assert plan.stats()["general_queries"] > 0
            for a, b in zip(got, want):
                assert torch.equal(a, b), (scout, most, workers)
    monkeypatch.delenv("HGX_LIFT_SCOUT")
    monkeypatch.delenv("HGX_LIFT_SCOUT_MAX")
    monkeypatch.delenv("HGX_LIFT_WORKERS")
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    for _ in range(3):  # (the first run has no count to go by; the later ones do)
        for a, b in zip(run(plan), want):
            assert torch.equal(a, b)
    # batches of changing sizes through one plan: the two sets of words the scouts count in alternate by batch, and a batch's tiles
    # clear the other set only as far as the batch goes — a longer batch behind a shorter one finds words the plan has to clear
    sizes = (70000, 20000, 70000, 33000, 17000, 70000)

    def run_sizes(p):
        out = []
        for k, m in enumerate(sizes):
            gs, ge, st = batches[k % 2]
            ptr, nrec = p.run(gs[:m].contiguous(), ge[:m].contiguous(), st[:m].contiguous())
            out.append(p.records_to_tensor(ptr, nrec).cpu())
        return out

    inline = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    inline.set_workers(0)
    want_sizes = run_sizes(inline)
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    run(plan)
    for a, b in zip(run_sizes(plan), want_sizes):
        assert torch.equal(a, b)
