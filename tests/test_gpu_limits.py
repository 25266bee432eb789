"""Shapes that used to hit fixed limits of the device structures: a genome with more than 16 children (star trees, pangenome
roots) and alignments of more than 256 genomes (a 241-way alignment has 481 with its ancestors).  Liftover, depth and MAF
against the oracle."""
import subprocess

import pytest

import halfix
from util import oracle_liftover
from test_gpu_multiseq import _bed

pytestmark = pytest.mark.gpu


def _oracle(oracle_bin, cmd, img, tmp_path, *args):
    out = str(tmp_path / ("o." + cmd))
    if cmd == "maf":
        subprocess.check_call([oracle_bin, "maf", img, out] + list(args))
    else:
        subprocess.check_call([oracle_bin, "depth", img, args[0], out] + list(args[1:]))
    return open(out).read()


def _check(hal, oracle_bin, tmp_path, al, img, pairs, refs):
    for s, t in pairs:
        bed = _bed(al, s, 80, 11 * s + t)
        for forced in ("1", "0"):
            import os
            os.environ["HGX_COMPOSED_UP"] = forced
            try:
                assert hal.liftover_convert(al, s, bed, t) == oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path), \
                    (al.genome_name(s), al.genome_name(t), forced)
            finally:
                del os.environ["HGX_COMPOSED_UP"]
    for g in refs:
        name = al.genome_name(g)
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        assert al.alignment_depth(g, count_dupes=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--countDupes"), name
        assert al.maf_export(g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), name


def test_star_tree_with_twentyfour_children(hal, oracle_bin, tmp_path):
    img = str(tmp_path / "star.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(5, n_genomes=41, max_children=3, root_len=500, root_children=24))
    al = hal.Alignment.open(img, device=0)
    assert max(len(al.genome_children(g)) for g in range(al.num_genomes)) > 16
    n = al.num_genomes
    _check(hal, oracle_bin, tmp_path, al, img, [(1, 40), (40, 1), (0, 17), (23, 0), (9, 31)], [0, 1, n - 1])


def test_three_hundred_genomes(hal, oracle_bin, tmp_path):
    img = str(tmp_path / "wide.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(7, n_genomes=300, max_children=3, root_len=300))
    al = hal.Alignment.open(img, device=0)
    assert al.num_genomes == 300
    _check(hal, oracle_bin, tmp_path, al, img, [(299, 150), (150, 299), (0, 280), (270, 0)], [0, 299, 120])


def test_depth_sweeps_with_more_than_64_counted_genomes(hal, oracle_bin, tmp_path, monkeypatch):
    """the tree sweeps keep a 64-bit genome set per base: alignments with more counted genomes (300 here, 150 leaves with
    --noAncestors, a hundred targets) go through them in groups of 64, the groups' set sizes added up — against the column walk
    and the oracle"""
    import numpy as np
    img = str(tmp_path / "wide.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(7, n_genomes=300, max_children=3, root_len=300))
    al = hal.Alignment.open(img, device=0)
    leaves = [g for g in range(al.num_genomes) if not al.genome_children(g)]
    assert len(leaves) > 64
    for g in (0, 299, 120, leaves[0], leaves[-1]):
        name = al.genome_name(g)
        n = al.genome_length(g)
        if n == 0:
            continue
        leaf = not al.genome_children(g)
        cases = [dict(), dict(count_dupes=True), dict(targets=list(range(3, 290, 3)))]
        if leaf:
            cases.append(dict(no_ancestors=True))
        for kw in cases:
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
            a = al.columns_depth(g, 0, n, **kw)
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "0")
            assert np.array_equal(a, al.columns_depth(g, 0, n, **kw)), (name, kw)
        monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        if leaf:
            assert al.alignment_depth(g, no_ancestors=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--noAncestors"), name


def test_count_dupes_sweep_over_a_polytomy_with_segment_tails(hal, oracle_bin, tmp_path, monkeypatch):
    """--countDupes by the tree sweeps when a genome has more than eight children in scope (k_sweep_up runs once per eight and
    adds to what the launch before left) and bottom segments whose length leaves one base past a round of lanes (33, 65, 97):
    the lane at a segment's end must not add a launch's children twice to a base its neighbour of the round before has stored."""
    import random
    import numpy as np
    rnd = random.Random(3)
    lens = [33, 65, 33, 97, 1, 34, 65, 129, 2, 33]
    starts = [sum(lens[:i]) for i in range(len(lens))]
    total = sum(lens)
    nkids = 12
    genomes = [None] * (nkids + 1)
    slots = []
    for c in range(nkids):
        tops, pos = [], 0
        members = {}
        order = list(range(len(lens))) + [rnd.randrange(len(lens)) for _ in range(4)]  # (every segment once, four of them twice)
        rnd.shuffle(order)
        for k, j in enumerate(order):
            tops.append([pos, lens[j], j, rnd.random() < 0.4, -1])
            members.setdefault(j, []).append(k)
            pos += lens[j]
        for j, ms in members.items():
            if len(ms) > 1:
                for a, b in zip(ms, ms[1:] + ms[:1]):
                    tops[a][4] = b
        genomes[c + 1] = halfix.simple_genome("L%d" % c, 0, [], pos, [tuple(t) for t in tops], [], seqname="L%d_chr" % c)
        slots.append({j: (ms[-1], tops[ms[-1]][3]) for j, ms in members.items()})
    bots = [(starts[j], lens[j], [slots[c][j] for c in range(nkids)]) for j in range(len(lens))]
    genomes[0] = halfix.simple_genome("Root", -1, list(range(1, nkids + 1)), total, [], bots, seqname="Root_chr")
    img = str(tmp_path / "poly.hgx")
    halfix.write_hgx(img, genomes)
    al = hal.Alignment.open(img, device=0)
    for g in (0, 1, nkids):
        name, n = al.genome_name(g), al.genome_length(g)
        for kw in (dict(count_dupes=True), dict()):
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
            a = al.columns_depth(g, 0, n, **kw)
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "0")
            assert np.array_equal(a, al.columns_depth(g, 0, n, **kw)), (name, kw)
        monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
        assert al.alignment_depth(g, count_dupes=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--countDupes"), name
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
