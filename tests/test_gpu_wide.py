"""The int64-coordinate instantiation of every kernel (taken by genomes of 2^31 bases or more), forced on small
alignments with HGX_FORCE_WIDE=1 and checked against the oracle."""
import os

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
