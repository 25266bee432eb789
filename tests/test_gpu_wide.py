"""The int64-coordinate instantiation of every kernel (taken by genomes of 2^31 bases or more — hal_index_t is int64,
api/inc/halDefs.h:34), forced on small alignments with HGX_FORCE_WIDE=1 and taken by itself on alignments whose genomes have
more than 2^31 bases, checked against the oracle; the single-pass kernels over the merged table (composed_kind 3) included."""
import os

import numpy as np
import pytest

import halfix
import handbuilt_liftover as hb
from util import random_bed, oracle_liftover
from test_gpu_columns import _oracle

pytestmark = pytest.mark.gpu


@pytest.fixture()
def wide(monkeypatch):
    monkeypatch.setenv("HGX_FORCE_WIDE", "1")


def test_wide_liftover_and_columns(hal, oracle_bin, tmp_path, wide):
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=8,
                           max_segment_length=40, min_segments=300, max_segments=700, seed=2, with_dna=True)
    al = hal.Alignment.random(opts, device=0)
    img = str(tmp_path / "w.hgx")
    al.save(img)
    n = al.num_genomes
    lines = 0
    for s, t in [(n - 1, 2), (n - 1, n - 2), (0, n - 1), (n - 1, 0), (4, 4), (5, 7), (1, n - 1)]:
        name, _, length = al.sequences(s)[0]
        bed = random_bed(name, length, 400, 1, 600, s * 31 + t, strands="+-.")
        for nd in (False, True):
            got = hal.liftover_convert(al, s, bed, t, traverse_dupes=not nd)
            assert got == oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, no_dupes=nd), (s, t, nd)
            lines += got.count("\n")
    assert lines > 3000
    # wide fan-out (deferred finishing path) in the wide instantiation
    name, _, length = al.sequences(n - 1)[0]
    bed = random_bed(name, length, 30, 3000, 6000, 9)
    assert hal.liftover_convert(al, n - 1, bed, n - 2) == oracle_liftover(oracle_bin, img, al.genome_name(n - 1), al.genome_name(n - 2), bed, tmp_path)
    for g in (0, 3, n - 1):
        nm = al.genome_name(g)
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, nm)
        assert al.maf_export(g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm)


def test_wide_handbuilt_and_multiseq(hal, oracle_bin, tmp_path, wide):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    for src, tgt, bed, want in hb.CASES:
        assert hal.liftover_convert(al, al.genome_id(src), bed, al.genome_id(tgt)) == want
    for src, tgt, bed, want, psl, pslname in hb.CASES12:
        assert hal.liftover_convert(al, al.genome_id(src), bed, al.genome_id(tgt), out_psl=psl, out_psl_with_name=pslname) == want
    img2 = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img2, halfix.random_multiseq_alignment(4))
    al2 = hal.Alignment.open(img2, device=0)
    for g in range(al2.num_genomes):
        nm = al2.genome_name(g)
        assert al2.maf_export(g, unique=True) == _oracle(oracle_bin, "maf", img2, tmp_path, "--refGenome", nm, "--unique")
        for t in range(al2.num_genomes):
            seqs = al2.sequences(g)
            bed = "".join("%s\t0\t%d\tw\t0\t-\n" % (s[0], s[2]) for s in seqs if s[2] > 0)
            assert hal.liftover_convert(al2, g, bed, t) == oracle_liftover(oracle_bin, img2, nm, al2.genome_name(t), bed, tmp_path)


def _records(hal, al, src, tgt, gs, ge, st, dupes=True):
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=int(gs.numel()), traverse_dupes=dupes)
    ptr, nrec = plan.run(gs, ge, st)
    return plan, plan.records_to_tensor(ptr, nrec).cpu()


def test_wide_single_pass_kernels(hal, oracle_bin, tmp_path, wide, monkeypatch):
    """HGX_FORCE_WIDE=1: the plan of an int64 alignment is served by k_lift_classify / k_lift_merged (composed_kind 3) and gives
    the oracle's lines for every genome pair, and the walk's records at scale — general intervals, submit / collect included."""
    import torch
    monkeypatch.setenv("HGX_COMPOSED_UP", "1")
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10,
                           max_segment_length=60, min_segments=200, max_segments=600, seed=5, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    img = str(tmp_path / "w.hgx")
    al.save(img)
    n = al.num_genomes
    lines = 0
    for s in range(n):
        name, _, length = al.sequences(s)[0]
        for t in range(n):
            bed = random_bed(name, length, 150, 1, 500, s * n + t, strands="+-.")
            nd = (s + t) % 3 == 0
            got = hal.liftover_convert(al, s, bed, t, traverse_dupes=not nd)
            assert got == oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, no_dupes=nd), (s, t, nd)
            lines += got.count("\n")
            m = al.mrca(s, t)
            if (s + 2 * t) % 5 == 0 and al.genome_parent(m) >= 0:
                lim = al.genome_parent(m)
                assert hal.liftover_convert(al, s, bed, t, coalescence_limit=lim) == \
                    oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, coalescence_limit=al.genome_name(lim))
    assert lines > 3000
    # at scale: the merged table against the walk, record for record; two batches in flight
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                           max_segment_length=200, min_segments=70000, max_segments=140000, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    _, ss, length = al.sequences(src)[0]
    nq = 200000
    g = torch.Generator().manual_seed(4)
    starts = torch.randint(0, length - 1100, (nq,), generator=g)
    lens = torch.randint(50, 1000, (nq,), generator=g)
    strand = torch.where(torch.rand(nq, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8)
    gs, ge, st = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
    plan, table = _records(hal, al, src, tgt, gs, ge, st)
    stats = plan.stats()
    assert stats["composed_kind"] == 3 and stats["general_queries"] > 0
    kt = plan.kernel_times()
    assert "k_lift_classify" in kt and "k_lift_merged" in kt and "k_finish_fast" not in kt
    monkeypatch.setenv("HGX_COMPOSED_UP", "0")
    wplan, walk = _records(hal, al, src, tgt, gs, ge, st)
    assert wplan.stats()["composed_kind"] == 0
    assert table.shape[0] > nq and torch.equal(table, walk)
    monkeypatch.setenv("HGX_COMPOSED_UP", "1")
    plans = [hal.LiftoverPlan(al, src, tgt, max_queries=nq) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    for p in plans:
        p.set_timing(0)
        p.run(gs, ge, st)  # (the first run builds nothing new: the alignment has its table)
    for rounds in range(3):
        for p, sm in zip(plans, streams):
            p.submit(gs, ge, st, stream=sm)
        for p in plans:
            ptr, nrec = p.collect()
            assert p.stats()["composed_kind"] == 3
            assert torch.equal(p.records_to_tensor(ptr, nrec).cpu(), walk)


def test_genomes_of_more_than_2_31_bases(hal, oracle_bin, tmp_path, monkeypatch):
    """Alignments whose coordinates do not fit 32 bits (the independent test generator's alignments scaled by 2 * 10^7: genomes of
    5-10 * 10^9 bases in a few hundred segments each): the int64 tables are chosen by the library itself; the merged table
    (junction and chain sorts in two passes), the single-pass kernels and the walk against the oracle."""
    K = 20_000_000
    for seed in (1, 4):
        img = str(tmp_path / ("big%d.hgx" % seed))
        halfix.write_hgx(img, halfix.scale_alignment(halfix.random_multiseq_alignment(seed, n_genomes=6), K))
        rng = np.random.default_rng(seed)
        for mode in ("1", "0"):
            monkeypatch.setenv("HGX_COMPOSED_UP", mode)
            al = hal.Alignment.open(img, device=0)
            assert max(al.genome_length(g) for g in range(al.num_genomes)) > 2 ** 32
            n = al.num_genomes
            for s in range(n):
                seqs = [q for q in al.sequences(s) if q[2] > 0]
                for t in range(n):
                    lines = []
                    for i in range(80):
                        name, _, length = seqs[int(rng.integers(0, len(seqs)))]
                        # short intervals around segment boundaries (multiples of K), long ones over several segments
                        if i % 2:
                            ln = int(rng.integers(1, 3 * K))
                            st = int(rng.integers(0, length - min(ln, length) + 1))
                        else:
                            edge = int(rng.integers(0, length // K + 1)) * K
                            st = max(0, min(length - 1, edge - int(rng.integers(0, 50))))
                            ln = int(rng.integers(1, 100))
                        en = min(length, st + ln)
                        lines.append("%s\t%d\t%d\tq%d\t0\t%s\n" % (name, st, en, i, "+-."[int(rng.integers(0, 3))]))
                    bed = "".join(lines)
                    nd = (s + t) % 2 == 1
                    got = hal.liftover_convert(al, s, bed, t, traverse_dupes=not nd)
                    assert got == oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, no_dupes=nd), \
                        (seed, mode, al.genome_name(s), al.genome_name(t), nd)
    # the device entry point on such an alignment: served by the merged table
    import torch
    monkeypatch.setenv("HGX_COMPOSED_UP", "1")
    al = hal.Alignment.open(img, device=0)
    n = al.num_genomes
    src, tgt = [(a, b) for a in range(n - 1, 0, -1) for b in range(1, n) if al.mrca(a, b) not in (a, b) and al.num_top_segments(a) > 0][0]
    _, seq_start, length = max(al.sequences(src), key=lambda q: q[2])  # (intervals stay inside one sequence, as BED lines do)
    g = torch.Generator().manual_seed(7)
    nq = 5000
    lens = torch.randint(1, min(2 * K, length // 2), (nq,), generator=g)
    starts = seq_start + (torch.rand(nq, generator=g, dtype=torch.float64) * (length - lens).to(torch.float64)).to(torch.int64)
    strand = torch.where(torch.rand(nq, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8)
    gs, ge, st = starts.cuda(), (starts + lens - 1).cuda(), strand.cuda()
    plan, table = _records(hal, al, src, tgt, gs, ge, st)
    assert plan.stats()["composed_kind"] == 3
    monkeypatch.setenv("HGX_COMPOSED_UP", "0")
    _, walk = _records(hal, al, src, tgt, gs, ge, st)
    assert table.shape[0] > 0 and torch.equal(table, walk)
