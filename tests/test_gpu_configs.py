"""BASELINE.json's configurations as parity tests of the HIP path (the bench times configs[1]; the others are cases):
  configs[1]  10-genome ~100 Mb/genome alignment at FULL size, Genome_9 -> Genome_2: 120 k intervals of the 1 M batch against
              the oracle (default plan: merged table + single-pass kernels), and the whole 1 M batch through the three plans;
  configs[2]  hal2maf on the same full-size alignment: a slice against the oracle, and the full-size properties;
  configs[3]  50-genome alignment, Genome_44 -> Genome_2: at a size the oracle covers with every option, and a full-size shard
              (1.25 M intervals) through the three plans;
  configs[4]  halAlignmentDepth of Genome_44 on the 50-genome alignment: against the oracle at reduced size, properties at full
              size (the two depth kernels agree; sharding the columns over ranks changes nothing).
Integer / byte work: the bar is bit-exact."""
import os
import subprocess
import sys

import numpy as np
import pytest

from util import oracle_liftover, random_bed

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_queries, workload_options  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg2_full(hal):
    return hal.Alignment.random(workload_options(1.0, "cfg2", dna="fast"), device=0)


@pytest.fixture(scope="module")
def cfg4_full(hal):
    return hal.Alignment.random(workload_options(1.0, "cfg4"), device=0)


def _bed(seq_name, starts, lens, strand, lo, hi):
    return "".join("%s\t%d\t%d\tq\t0\t%s\n" % (seq_name, int(starts[i]), int(starts[i] + lens[i]), chr(int(strand[i]))) for i in range(lo, hi))


def _three_plans(hal, al, src, tgt, gs, ge, st, nq):
    """the batch through the default plan (table from the first batch on), the unmerged table and the level walk"""
    import torch
    out = {}
    for name, env in (("default", {}), ("through", {"HGX_COMPOSED_UP": "1", "HGX_MERGED": "0"}), ("walk", {"HGX_COMPOSED_UP": "0"})):
        for k in ("HGX_COMPOSED_UP", "HGX_MERGED"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            if name == "through":  # tables are cached per alignment and genome pair: the unmerged one needs its own image
                al2 = hal.Alignment.random(workload_options(1.0, "cfg2" if al.num_genomes == 10 else "cfg4"), device=0)
            else:
                al2 = al
            plan = hal.LiftoverPlan(al2, src, tgt, max_queries=nq)
            ptr, nrec = plan.run(gs, ge, st)
            out[name] = (plan.records_to_tensor(ptr, nrec).cpu(), plan.stats())
            del plan
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert out["default"][1]["composed_kind"] == 3 and out["through"][1]["composed_kind"] == 2 and out["walk"][1]["composed_kind"] == 0
    assert torch.equal(out["default"][0], out["walk"][0]) and torch.equal(out["through"][0], out["walk"][0])
    return out["default"][0]


def test_config2_full_size_sample_vs_oracle(hal, oracle_bin, tmp_path, cfg2_full):
    al = cfg2_full
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    seq_name, seq_start, length = al.sequences(src)[0]
    assert length > 50_000_000 and al.num_genomes == 10
    starts, lens, strand = make_queries(length, 1000000, 1234)
    n = 120000
    bed = _bed(seq_name, starts, lens, strand, 0, n)
    img = str(tmp_path / "cfg2.hgx")
    al.save(img)
    got = hal.liftover_convert(al, src, bed, tgt)
    assert got == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", bed, tmp_path)
    assert got.count("\n") > 2 * n
    # the second pair SURVEY 8(d) names (3 hops up, 2 down), dupes off, '.' strands
    bed2 = _bed(seq_name, starts, lens, strand, n, n + 30000).replace("\t+\n", "\t.\n")
    t8 = al.genome_id("Genome_8")
    assert hal.liftover_convert(al, src, bed2, t8, traverse_dupes=False) == \
        oracle_liftover(oracle_bin, img, "Genome_9", "Genome_8", bed2, tmp_path, no_dupes=True)
    # configs[2] on the same image: hal2maf --refGenome Genome_9 --noAncestors over a slice, and halAlignmentDepth of it
    a, ln = length // 3, 200000
    maf = str(tmp_path / "o.maf")
    subprocess.check_call([oracle_bin, "maf", img, maf, "--refGenome", "Genome_9", "--noAncestors", "--refSequence", seq_name, "--start", str(a),
                           "--length", str(ln)])
    assert al.maf_export(src, 0, start=a, length=ln, no_ancestors=True) == open(maf).read()
    wig = str(tmp_path / "o.wig")
    subprocess.check_call([oracle_bin, "depth", img, "Genome_9", wig, "--refSequence", seq_name, "--start", str(a), "--length", str(ln)])
    assert al.alignment_depth(src, 0, start=a, length=ln) == open(wig).read()


def test_config2_full_batch_through_the_three_plans(hal, cfg2_full):
    """1 M intervals: the records of the default plan, of the unmerged table and of the level walk are the same bytes; and the
    size-independent properties of the output (grouped by interval in input order, sorted by source start inside an interval,
    inside the target sequence)."""
    al = cfg2_full
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    _, seq_start, length = al.sequences(src)[0]
    nq = 1000000
    starts, lens, strand = make_queries(length, nq, 1234)
    gs, ge, st = (starts + seq_start).cuda(), (starts + lens - 1 + seq_start).cuda(), strand.cuda()
    recs = _three_plans(hal, al, src, tgt, gs, ge, st, nq).numpy().view(hal.RECORD_DTYPE).reshape(-1)
    assert len(recs) > 3 * nq
    q = recs["query"]
    assert np.all(np.diff(q) >= 0) and q[0] >= 0 and q[-1] < nq
    same = np.diff(q) == 0
    assert np.all(np.diff(recs["src_start"])[same] >= 0)
    tlen = al.sequences(tgt)[0][2]
    assert recs["tgt_start"].min() >= 0 and recs["tgt_end"].max() <= tlen and np.all(recs["tgt_end"] > recs["tgt_start"])
    lo, hi = starts.numpy()[q] + seq_start, (starts + lens).numpy()[q] + seq_start
    assert np.all(recs["src_start"] >= lo) and np.all(recs["src_start"] < hi)
    assert np.all((recs["tgt_end"] - recs["tgt_start"]) <= lens.numpy()[q])


def test_config2_full_batch_scouts_list_and_inline(hal, cfg2_full, monkeypatch):
    """round 6: the 1 M-interval batch through a plan whose general intervals are finished by scouts (the default for a batch that
    runs by itself, from the second run on), by round 3's listed workers (HGX_LIFT_SCOUT=0) and inline (no workers): the same bytes;
    and through two plans with a batch each in flight (inline there)."""
    import torch
    al = cfg2_full
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    _, seq_start, length = al.sequences(src)[0]
    nq = 1000000
    batches = []
    for seed in (1234, 77):
        starts, lens, strand = make_queries(length, nq, seed)
        batches.append(((starts + seq_start).cuda(), (starts + lens - 1 + seq_start).cuda(), strand.cuda()))

    def three_runs(plan):
        out = None
        for k in range(3):  # (the first run has no count of general intervals to go by: the later ones take the workers)
            out = []
            for b in batches:
                ptr, nrec = plan.run(*b)
                out.append(plan.records_to_tensor(ptr, nrec).cpu())
        return out

    inline = hal.LiftoverPlan(al, src, tgt, max_queries=nq)
    inline.set_workers(0)
    want = three_runs(inline)
    assert inline.stats()["general_queries"] > 500
    for scout in ("1", "0"):
        monkeypatch.setenv("HGX_LIFT_SCOUT", scout)
        plan = hal.LiftoverPlan(al, src, tgt, max_queries=nq)
        got = three_runs(plan)
        for a, b in zip(got, want):
            assert torch.equal(a, b), scout
    monkeypatch.delenv("HGX_LIFT_SCOUT")
    plans = [hal.LiftoverPlan(al, src, tgt, max_queries=nq) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for p in plans:
        for b in batches:
            p.run(*b)
    pending = [None, None]  # (the batch a plan has in flight)
    for i in range(6):
        k = i & 1
        if pending[k] is not None:
            ptr, nrec = plans[k].collect()
            with torch.cuda.stream(streams[k]):
                assert torch.equal(plans[k].records_to_tensor(ptr, nrec).cpu(), want[pending[k]]), i
        pending[k] = (i >> 1) & 1
        plans[k].submit(*batches[pending[k]], stream=streams[k])
    for k in (0, 1):
        ptr, nrec = plans[k].collect()
        with torch.cuda.stream(streams[k]):
            assert torch.equal(plans[k].records_to_tensor(ptr, nrec).cpu(), want[pending[k]])


def test_config3_full_size_maf_properties(hal, cfg2_full):
    """hal2maf --refGenome Genome_9 --noAncestors over 3 M columns of the full-size alignment: every reference base is in exactly
    one block (the reference rows tile the range), rows of a block have equal text length, and the export of the range equals
    the concatenation of its halves' blocks (MafExport starts a fresh block at a range start: maf/hal2mafMP.py:63-79)."""
    al = cfg2_full
    src = al.genome_id("Genome_9")
    seq_name = al.sequences(src)[0][0]
    a, ln = 7_000_000, 3_000_000
    text = al.maf_export(src, 0, start=a, length=ln, no_ancestors=True)
    pos, blocks = a, 0
    width = None
    for line in text.split("\n"):
        if line.startswith("a"):
            blocks += 1
            width = None
        elif line.startswith("s\t"):
            f = line.split("\t")
            if width is None:  # the reference row comes first
                assert f[1] == "Genome_9." + seq_name and f[4] == "+" and int(f[2]) == pos
                pos += int(f[3])
                width = len(f[6])
            assert len(f[6]) == width
    assert pos == a + ln and blocks > ln // 200
    body = lambda t: t.split("\n\n", 1)[1]  # noqa: E731  (without the header)
    h1 = al.maf_export(src, 0, start=a, length=ln // 2, no_ancestors=True)
    h2 = al.maf_export(src, 0, start=a + ln // 2, length=ln - ln // 2, no_ancestors=True)
    whole = body(text).split("\n\n")
    halves = body(h1).rstrip("\n").split("\n\n") + body(h2).rstrip("\n").split("\n\n")
    # the halves cut one block in two at the seam; every other block is the same
    assert len(halves) in (len([b for b in whole if b.strip()]), len([b for b in whole if b.strip()]) + 1)
    assert halves[:50] == whole[:50]


def test_config4_and_5_against_the_oracle(hal, oracle_bin, tmp_path):
    """The 50-genome alignment (seed 0, mean degree 2) at a size the oracle covers: Genome_44 -> Genome_2 (7 hops up, 1 down)
    with every liftover option, and halAlignmentDepth of Genome_44."""
    al = hal.Alignment.random(workload_options(0.004, "cfg4"), device=0)
    assert al.num_genomes == 50
    img = str(tmp_path / "cfg4.hgx")
    al.save(img)
    src, tgt = al.genome_id("Genome_44"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    for forced in ("1", "0"):  # tables (merged, single-pass kernels) and level walk
        os.environ["HGX_COMPOSED_UP"] = forced
        try:
            bed = random_bed(name, length, 6000, 50, 1000, 44, strands="+-.")
            assert hal.liftover_convert(al, src, bed, tgt) == oracle_liftover(oracle_bin, img, "Genome_44", "Genome_2", bed, tmp_path)
            bed = random_bed(name, length, 2000, 1, 3000, 45)
            assert hal.liftover_convert(al, src, bed, tgt, traverse_dupes=False) == \
                oracle_liftover(oracle_bin, img, "Genome_44", "Genome_2", bed, tmp_path, no_dupes=True)
            lim = al.genome_parent(al.mrca(src, tgt))
            if lim >= 0:
                assert hal.liftover_convert(al, src, bed, tgt, coalescence_limit=lim) == \
                    oracle_liftover(oracle_bin, img, "Genome_44", "Genome_2", bed, tmp_path, coalescence_limit=al.genome_name(lim))
            other = al.genome_id("Genome_30")
            assert hal.liftover_convert(al, src, bed, other) == oracle_liftover(oracle_bin, img, "Genome_44", "Genome_30", bed, tmp_path)
        finally:
            del os.environ["HGX_COMPOSED_UP"]
    wig = str(tmp_path / "o.wig")
    subprocess.check_call([oracle_bin, "depth", img, "Genome_44", wig])
    assert al.alignment_depth(src) == open(wig).read()
    subprocess.check_call([oracle_bin, "depth", img, "Genome_44", wig, "--countDupes"])
    assert al.alignment_depth(src, count_dupes=True) == open(wig).read()


def test_config4_full_size_shard_through_the_three_plans(hal, cfg4_full):
    al = cfg4_full
    src, tgt = al.genome_id("Genome_44"), al.genome_id("Genome_2")
    _, seq_start, length = al.sequences(src)[0]
    nq = 1250000  # one GPU's shard of the 10 M intervals
    starts, lens, strand = make_queries(length, nq, 1234)
    gs, ge, st = (starts + seq_start).cuda(), (starts + lens - 1 + seq_start).cuda(), strand.cuda()
    recs = _three_plans(hal, al, src, tgt, gs, ge, st, nq).numpy().view(hal.RECORD_DTYPE).reshape(-1)
    q = recs["query"]
    assert len(recs) > 10 * nq and np.all(np.diff(q) >= 0)
    assert np.all(np.diff(recs["src_start"])[np.diff(q) == 0] >= 0)


def test_config5_full_size_depth_properties(hal, cfg4_full):
    """Whole-genome depth of Genome_44 on the 50-genome alignment: the run kernel and the per-column kernel agree on a 4 M-column
    window, the ranks' shards of the column range concatenate to the unsharded result (hal_amd.shard.shard_bounds), bounds."""
    import torch
    from hal_amd import shard
    al = cfg4_full
    src = al.genome_id("Genome_44")
    ncol = al.genome_length(src)
    whole = torch.empty(ncol, dtype=torch.int32, device="cuda")
    al.columns_depth_device(src, 0, ncol, whole.data_ptr())
    assert int(whole.min()) >= 0 and int(whole.max()) <= al.num_genomes - 1 and int(whole.max()) > 5
    parts = []
    for r in range(8):
        lo, hi = shard.shard_bounds(ncol, 8, r)
        part = torch.empty(hi - lo, dtype=torch.int32, device="cuda")
        al.columns_depth_device(src, lo, hi - lo, part.data_ptr())
        parts.append(part)
    assert torch.equal(torch.cat(parts), whole)
    n = 4_000_000
    os.environ["HGX_COLUMNS_PER_BASE"] = "1"
    try:
        win = torch.empty(n, dtype=torch.int32, device="cuda")
        al.columns_depth_device(src, ncol // 2, n, win.data_ptr())
    finally:
        del os.environ["HGX_COLUMNS_PER_BASE"]
    assert torch.equal(win, whole[ncol // 2:ncol // 2 + n])
