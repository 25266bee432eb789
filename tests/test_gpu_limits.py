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
