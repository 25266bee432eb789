"""Parity of the HIP liftover path (through the C ABI) with the oracle and the reference's golden vectors.
Integer/byte work: the bar is bit-exact text."""
import bz2
import os

import numpy as np
import pytest

import halfix
import handbuilt_liftover as hb
from util import random_bed, oracle_liftover

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rand_alignment(hal, tmp_path, seed, max_branch=3.0, min_seg=10, max_seg=60, min_segs=200, max_segs=600, max_genomes=10,
                    mean_degree=1.5, name="rnd", with_dna=False):
    opts = hal.RandOptions(mean_degree=mean_degree, max_branch_length=max_branch, min_genomes=2, max_genomes=max_genomes,
                           min_segment_length=min_seg, max_segment_length=max_seg, min_segments=min_segs,
                           max_segments=max_segs, seed=seed, with_dna=with_dna)
    al = hal.Alignment.random(opts, device=0)
    img = str(tmp_path / ("%s_%d.hgx" % (name, seed)))
    al.save(img)
    return al, img


def _check_pair(hal, oracle_bin, al, img, src, tgt, tmp_path, n=300, min_len=1, max_len=300, seed=0, no_dupes=False,
                strands="+-", bed6=True):
    name, _, length = al.sequences(src)[0]
    if length == 0:
        return 0
    bed = random_bed(name, length, n, min_len, max_len, seed, strands=strands, bed6=bed6)
    got = hal.liftover_convert(al, src, bed, tgt, traverse_dupes=not no_dupes)
    want = oracle_liftover(oracle_bin, img, al.genome_name(src), al.genome_name(tgt), bed, tmp_path, no_dupes=no_dupes)
    assert got == want, "src=%s tgt=%s noDupes=%s" % (al.genome_name(src), al.genome_name(tgt), no_dupes)
    return got.count("\n")


def test_reference_unit_test_handbuilt(hal, tmp_path):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    for src, tgt, bed, want in hb.CASES:
        assert hal.liftover_convert(al, al.genome_id(src), bed, al.genome_id(tgt)) == want, (src, tgt)


def test_handbuilt_all_pairs_vs_oracle(hal, oracle_bin, tmp_path):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    total = 0
    for s in range(al.num_genomes):
        for t in range(al.num_genomes):
            for nd in (False, True):
                total += _check_pair(hal, oracle_bin, al, img, s, t, tmp_path, n=200, max_len=100, seed=s * 7 + t, no_dupes=nd,
                                     strands="+-.")
    assert total > 1000


def test_reference_cli_goldens(hal, tmp_path):
    al = hal.Alignment.random(hal.RandOptions.preset("small", seed=0), device=0)
    g0, g2 = al.genome_id("Genome_0"), al.genome_id("Genome_2")
    bed = open(os.path.join(GOLD, "ref_liftover", "test1.bed3")).read()
    assert hal.liftover_convert(al, g0, bed, g2) == open(os.path.join(GOLD, "ref_liftover", "halLiftoverBed3Test.bed")).read()
    bed = open(os.path.join(GOLD, "ref_liftover", "test1.bed4+2")).read()
    assert hal.liftover_convert(al, g0, bed, g2, bed_type=4) == \
        open(os.path.join(GOLD, "ref_liftover", "halLiftoverBed4ExtraTest.bed")).read()


def test_reference_mmap_file_all_pairs(hal, oracle_bin, tmp_path):
    raw = bz2.decompress(open(os.path.join(GOLD, "ref_mmap", "small.mmap1.0.hal.bz2"), "rb").read())
    p = tmp_path / "small.mmap1.0.hal"
    p.write_bytes(raw)
    al = hal.Alignment.open(str(p), device=0)
    img = str(tmp_path / "mm.hgx")
    al.save(img)
    for s in range(al.num_genomes):
        for t in range(al.num_genomes):
            _check_pair(hal, oracle_bin, al, img, s, t, tmp_path, n=200, max_len=6000, seed=s + 10 * t)


@pytest.mark.parametrize("seed", [2, 5, 6])
def test_random_alignments_all_pairs(hal, oracle_bin, tmp_path, seed):
    """Inversions, transpositions, paralogy rings, incommensurate tilings (maxBranchLength 3)."""
    al, img = _rand_alignment(hal, tmp_path, seed)
    n = al.num_genomes
    lines = 0
    for s in range(n):
        for t in range(n):
            lines += _check_pair(hal, oracle_bin, al, img, s, t, tmp_path, n=150, max_len=250, seed=seed * 100 + s * n + t,
                                 no_dupes=((s + t) % 3 == 0), strands="+-.")
    assert lines > 5000


def test_wide_fanout_deferred_path(hal, oracle_bin, tmp_path):
    """Intervals spanning hundreds of tiny segments: pieces per interval exceed the LDS staging capacity, so
    the global-scratch finishing path runs."""
    al, img = _rand_alignment(hal, tmp_path, 2, min_seg=4, max_seg=12, min_segs=3000, max_segs=5000, name="tiny")
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_8")
    name, _, length = al.sequences(src)[0]
    bed = random_bed(name, length, 60, 2000, 6000, 3)
    got = hal.liftover_convert(al, src, bed, tgt)
    want = oracle_liftover(oracle_bin, img, "Genome_9", "Genome_8", bed, tmp_path)
    assert got == want and got.count("\n") > 1000
    # and down a long branch from the root with duplications
    src, tgt = al.genome_id("Genome_0"), al.genome_id("Genome_9")
    name, _, length = al.sequences(src)[0]
    bed = random_bed(name, length, 40, 3000, 8000, 4)
    assert hal.liftover_convert(al, src, bed, tgt) == oracle_liftover(oracle_bin, img, "Genome_0", "Genome_9", bed, tmp_path)


def test_overlap_breaking_paralogous_sources(hal, oracle_bin, tmp_path):
    """Long intervals on a heavily duplicated child: several source segments of one interval share ancestors, so the
    mapped set needs insertAndBreakOverlaps' refinement and extractSegment's equivalence classes."""
    al, img = _rand_alignment(hal, tmp_path, 6, max_branch=3.0, min_seg=5, max_seg=9, min_segs=40, max_segs=60, name="dup")
    hits = 0
    for s in range(al.num_genomes):
        for t in range(al.num_genomes):
            name, _, length = al.sequences(s)[0]
            if length == 0:
                continue
            bed = "".join("%s\t%d\t%d\tw%d\t0\t%s\n" % (name, a, min(length, a + w), a, st)
                          for a in range(0, length - 1, max(1, length // 7)) for w in (length, length // 2, 37)
                          for st in "+-")
            got = hal.liftover_convert(al, s, bed, t)
            want = oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path)
            assert got == want, (al.genome_name(s), al.genome_name(t))
            hits += got.count("\n")
    assert hits > 2000


def test_bed_columns_and_skips(hal, oracle_bin, tmp_path):
    al, img = _rand_alignment(hal, tmp_path, 5)
    src, tgt = al.genome_id("Genome_4"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    bed = ("%s\t10\t200\n" % name +                                            # BED3
           "\n  \n" +                                                         # blank lines are skipped
           "%s\t10\t200\tn\t5\t-\t10\t200\t1,2,3\n" % name +                   # BED9, thick fields rewritten
           "%s\t50\t90\tn2\t7\t+\t0\t0\t9\n" % name +                          # BED9, thick 0/0 kept, rgb "9" -> 9,9,9
           "nosuch\t1\t5\tx\t0\t+\n" +                                         # unknown sequence: skipped
           "%s\t5\t%d\ty\t0\t+\n" % (name, length + 10) +                      # end beyond the sequence: skipped
           "%s\t300\t420\tshort\n" % name)                                     # BED4 inherits the previous strand
    got = hal.liftover_convert(al, src, bed, tgt)
    want = oracle_liftover(oracle_bin, img, "Genome_4", "Genome_2", bed, tmp_path)
    assert got == want and got.count("\n") > 3
    # explicit --bedType with pass-through columns (liftover/Makefile:59-61)
    bed = "%s\t50\t290\tn2\t7\t-\textraA\textraB\n%s\t1\t99\tn3\t8\t.\tC\n" % (name, name)
    got = hal.liftover_convert(al, src, bed, tgt, bed_type=6)
    assert got == oracle_liftover(oracle_bin, img, "Genome_4", "Genome_2", bed, tmp_path, bed_type=6) and "extraB" in got


def test_malformed_line_reports_like_reference(hal, tmp_path):
    al, _ = _rand_alignment(hal, tmp_path, 5)
    src, tgt = al.genome_id("Genome_4"), al.genome_id("Genome_2")
    name = al.sequences(src)[0][0]
    good = "%s\t10\t200\ta\t0\t+\n" % name
    with pytest.raises(hal.HgxError, match=r"Error zero or negative length BED range: .* in input bed line 2") as ei:
        hal.liftover_convert(al, src, good + "%s\t50\t50\tb\t0\t+\n" % name + good, tgt)
    assert ei.value.partial_output == hal.liftover_convert(al, src, good, tgt)  # line 1 was already written
    with pytest.raises(hal.HgxError, match="Expected at least three columns in BED record"):
        hal.liftover_convert(al, src, "%s\t5\n" % name, tgt)
    with pytest.raises(hal.HgxError, match="Strand character must be"):
        hal.liftover_convert(al, src, "%s\t5\t9\tn\t0\t*\n" % name, tgt)


def test_batch_records_grouped_in_input_order(hal, oracle_bin, tmp_path):
    al, img = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(7)
    n = 500
    iv = np.zeros(n, dtype=hal.api.INTERVAL_DTYPE)
    iv["start"] = rng.integers(0, length - 300, n)
    iv["end"] = iv["start"] + rng.integers(1, 300, n)
    iv["strand"] = rng.choice([b"+", b"-"], n)
    iv["end"][5] = length + 1  # skipped like halLiftover.cpp:62-66
    recs = al.liftover_batch(src, tgt, iv)
    assert np.all(np.diff(recs["query"]) >= 0) and 5 not in set(recs["query"])
    for q in np.unique(recs["query"]):
        r = recs[recs["query"] == q]
        assert np.all(np.diff(r["src_start"]) >= 0)  # Liftover::visitLine's sort by source start
    tname = al.sequences(tgt)[0][0]
    bed = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (name, a["start"], a["end"], a["strand"].decode()) for a in iv if a["end"] <= length)
    text = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (tname, r["tgt_start"], r["tgt_end"], r["strand"].decode()) for r in recs)
    assert text == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", bed, tmp_path)


def test_device_resident_plan_and_stats(hal, oracle_bin, tmp_path, monkeypatch):
    monkeypatch.setenv("HGX_COMPOSED_UP", "0")  # this test is about the walk kernels (the composed table has its own file)
    import torch
    al, img = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_8")
    name, sstart, length = al.sequences(src)[0]
    n = 4000
    g = torch.Generator().manual_seed(1)
    starts = torch.randint(0, length - 400, (n,), generator=g)
    lens = torch.randint(1, 400, (n,), generator=g)
    strand = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8)
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    for _ in range(2):  # plans are reusable; results identical
        ptr, nrec = plan.run((starts + sstart).cuda(), (starts + lens - 1 + sstart).cuda(), strand.cuda())
        recs = plan.records_to_tensor(ptr, nrec).cpu().numpy().view(hal.RECORD_DTYPE).reshape(-1)
        tname = al.sequences(tgt)[0][0]
        bed = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (name, int(s), int(s + l), chr(int(c))) for s, l, c in zip(starts, lens, strand))
        text = "".join("%s\t%d\t%d\tq\t0\t%s\n" % (tname, r["tgt_start"], r["tgt_end"], r["strand"].decode()) for r in recs)
        want, stats = oracle_liftover(oracle_bin, img, "Genome_9", "Genome_8", bed, tmp_path, stats=True)
        assert text == want
    st = plan.stats()
    assert st["queries"] == n and st["records"] == nrec and st["mapped_pieces"] >= nrec
    assert st["top_derefs"] > 0 and st["bottom_derefs"] > 0 and st["total_ms"] >= st["walk_ms"] > 0
    kt = plan.kernel_times()
    assert "k_up_chain" in kt and "k_finish_fast" in kt
    # the chained up kernel against one launch per level: same records, same logical dereference counts
    os.environ["HGX_LEVEL_SYNC_UP"] = "1"
    try:
        plan2 = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    finally:
        del os.environ["HGX_LEVEL_SYNC_UP"]
    ptr2, nrec2 = plan2.run((starts + sstart).cuda(), (starts + lens - 1 + sstart).cuda(), strand.cuda())
    recs2 = plan2.records_to_tensor(ptr2, nrec2).cpu().numpy().view(hal.RECORD_DTYPE).reshape(-1)
    assert nrec2 == nrec and recs2.tobytes() == recs.tobytes()
    st2 = plan2.stats()
    assert "k_up_walk" in plan2.kernel_times()
    assert st2["top_derefs"] == st["top_derefs"] and st2["bottom_derefs"] == st["bottom_derefs"]


def test_strand_symmetry_property_at_scale(hal, tmp_path):
    """Size-independent property: lifting an interval on '-' gives the same target intervals as on '+', strands
    flipped, in the same order (the source flip is undone by the strand-aware merge).  Run at 200k intervals on a
    10-genome alignment of ~10 Mb genomes, too big for the oracle to check in seconds."""
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                           max_segment_length=200, min_segments=70000, max_segments=140000, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    name, _, length = al.sequences(src)[0]
    rng = np.random.default_rng(11)
    n = 200000
    iv = np.zeros(n, dtype=hal.api.INTERVAL_DTYPE)
    iv["start"] = rng.integers(0, length - 1000, n)
    iv["end"] = iv["start"] + rng.integers(50, 1000, n)
    iv["strand"] = b"+"
    plus = al.liftover_batch(src, tgt, iv)
    again = al.liftover_batch(src, tgt, iv)
    assert plus.tobytes() == again.tobytes()  # deterministic
    iv["strand"] = b"-"
    minus = al.liftover_batch(src, tgt, iv)
    assert len(plus) == len(minus) > n
    for f in ("query", "tgt_start", "tgt_end", "src_start", "tgt_seq"):
        assert np.array_equal(plus[f], minus[f]), f
    assert np.all((plus["strand"] == b"+") == (minus["strand"] == b"-"))
    # every record lies inside the target genome and has positive length
    tlen = al.sequences(tgt)[0][2]
    assert plus["tgt_start"].min() >= 0 and plus["tgt_end"].max() <= tlen and np.all(plus["tgt_end"] > plus["tgt_start"])
    # merged lines never exceed the query length
    qlen = (iv["end"] - iv["start"])[plus["query"]]
    assert np.all(plus["tgt_end"] - plus["tgt_start"] <= qlen)


def test_bed12_and_psl_reference_goldens(hal, tmp_path):
    """liftover/Makefile:38-57 goldens and the BED12 / PSL literal strings of halLiftoverTests.cpp:345-373."""
    al = hal.Alignment.random(hal.RandOptions.preset("small", seed=0), device=0)
    g0, g2 = al.genome_id("Genome_0"), al.genome_id("Genome_2")
    d = os.path.join(GOLD, "ref_liftover")
    rd = lambda n: open(os.path.join(d, n)).read()
    assert hal.liftover_convert(al, g0, rd("test1.bed12"), g2) == rd("halLiftoverBed12Test.bed")
    assert hal.liftover_convert(al, g0, rd("test1.bed12+2"), g2) == rd("halLiftoverBed12ExtraTest.bed")
    assert hal.liftover_convert(al, g0, rd("test1.bed12"), g2, out_psl=True) == rd("halLiftoverPsl12Test.psl")
    assert hal.liftover_convert(al, g0, rd("test1.bed3"), g2, out_psl=True) == rd("halLiftoverPsl3Test.psl")
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    hand = hal.Alignment.open(img, device=0)
    for src, tgt, bed, want, psl, pslname in hb.CASES12:
        got = hal.liftover_convert(hand, hand.genome_id(src), bed, hand.genome_id(tgt), out_psl=psl, out_psl_with_name=pslname)
        assert got == want, (psl, pslname)


def _random_bed12(seq_name, seq_len, n, seed, strands="+-"):
    rng = np.random.default_rng(seed)
    lines = []
    for i in range(n):
        nb = int(rng.integers(1, 6))
        span = int(rng.integers(nb * 4, min(seq_len, 900)))
        start = int(rng.integers(0, seq_len - span))
        cuts = np.sort(rng.choice(np.arange(1, span), size=2 * nb - 1, replace=False)) if span > 2 * nb else np.arange(1, 2 * nb)
        edges = [0] + list(map(int, cuts)) + [span]
        sizes = [edges[2 * k + 1] - edges[2 * k] for k in range(nb)]
        starts = [edges[2 * k] for k in range(nb)]
        order = rng.permutation(nb)  # blocks need not be sorted in the input
        st = strands[int(rng.integers(0, len(strands)))]
        lines.append("%s\t%d\t%d\tt%d\t0\t%s\t%d\t%d\t1,2,3\t%d\t%s\t%s\n" % (
            seq_name, start, start + span, i, st, start, start + span, nb, ",".join(str(sizes[k]) for k in order) + ",",
            ",".join(str(starts[k]) for k in order) + ","))
    return "".join(lines)


@pytest.mark.parametrize("seed", [2, 6])
def test_bed12_and_psl_vs_oracle(hal, oracle_bin, tmp_path, seed):
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10,
                           max_segment_length=60, min_segments=200, max_segments=600, seed=seed, with_dna=True)
    al = hal.Alignment.random(opts, device=0)
    img = str(tmp_path / "d.hgx")
    al.save(img)
    n = al.num_genomes
    pairs = [(n - 1, 2), (n - 1, n - 2), (0, n - 1), (n - 1, 0), (1, 1), (3, n - 1)]
    total = 0
    for s, t in pairs:
        name, _, length = al.sequences(s)[0]
        bed = _random_bed12(name, length, 120, seed * 10 + s, strands="+-.")
        for psl, pslname in ((False, False), (True, False), (True, True)):
            if psl:
                bed_in = bed.replace("\t.\t", "\t+\t")
            else:
                bed_in = bed
            got = hal.liftover_convert(al, s, bed_in, t, out_psl=psl, out_psl_with_name=pslname)
            want = oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed_in, tmp_path, psl=psl, psl_with_name=pslname)
            assert got == want, (al.genome_name(s), al.genome_name(t), psl, pslname)
            total += got.count("\n")
        # PSL from BED6 input (expandToBed12)
        bed6 = random_bed(name, length, 150, 5, 300, seed, strands="+-")
        assert hal.liftover_convert(al, s, bed6, t, out_psl=True) == \
            oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed6, tmp_path, psl=True)
    assert total > 500


def test_packed_wire_records_on_device(hal, tmp_path):
    """hal_amd.shard.pack_records / unpack_records on real device-resident records (the multi-GPU exchange format)."""
    import torch
    from hal_amd import shard
    al, _ = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    _, sstart, length = al.sequences(src)[0]
    n = 3000
    g = torch.Generator().manual_seed(3)
    starts = torch.randint(0, length - 300, (n,), generator=g)
    lens = torch.randint(1, 300, (n,), generator=g)
    strand = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8)
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    ptr, nrec = plan.run((starts + sstart).cuda(), (starts + lens - 1 + sstart).cuda(), strand.cuda())
    recs = plan.records_to_tensor(ptr, nrec)
    assert nrec > 1000 and shard.can_pack(max(al.genome_length(src), al.genome_length(tgt)), n, len(al.sequences(tgt)))
    packed = shard.pack_records(recs)
    assert packed.shape == (nrec, 20) and packed.is_cuda
    assert torch.equal(shard.unpack_records(packed), recs)
    # the library's own kernel writes the same bytes
    assert torch.equal(plan.records_to_tensor(ptr, nrec, packed=True), packed)


@pytest.mark.parametrize("forced", [None, "20", "40"])
def test_wire_blob_matches_the_reference_encoder(hal, tmp_path, monkeypatch, forced):
    """hgx_liftover_wire_blob (device kernels) writes byte for byte what hal_amd.shard.encode_blob builds from the plan's
    records, in every format, and decode_blob gives the records back with global query indices."""
    import torch
    from hal_amd import shard
    if forced:
        monkeypatch.setenv("HGX_WIRE_FORMAT", forced)
    al, _ = _rand_alignment(hal, tmp_path, 5)
    src, tgt = al.genome_id("Genome_3"), al.genome_id("Genome_1")
    _, ss, length = al.sequences(src)[0]
    n = 777
    g = torch.Generator().manual_seed(3)
    starts = torch.randint(0, length - 300, (n,), generator=g)
    lens = torch.randint(1, 300, (n,), generator=g)
    strand = torch.tensor([ord(c) for c in "+-."], dtype=torch.uint8)[torch.randint(0, 3, (n,), generator=g)]
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    ptr, nrec = plan.run((starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda())
    assert nrec > n // 2
    recs = plan.records_to_tensor(ptr, nrec)
    blob, fmt = plan.wire_blob(first_query=5000)
    assert fmt == int(forced or 12)
    assert torch.equal(blob.cpu(), shard.encode_blob(recs.cpu(), n, first_query=5000, fmt=fmt))
    out, first, nq = shard.decode_blob(blob)
    assert (first, nq) == (5000, n) and torch.equal(out, shard.offset_query_index(recs.clone(), 5000))
    assert blob.numel() == 32 + (((2 * n + 7) // 8 * 8 + 12 * nrec) if fmt == 12 else fmt * nrec)
