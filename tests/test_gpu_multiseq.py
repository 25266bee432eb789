"""Parity on alignments from the independent test generator (halfix.random_multiseq_alignment): several sequences per
genome, irregular segment lengths, insertions / deletions / inversions / duplications — structure halRandGen never
produces.  Liftover (BED6, BED12, PSL), depth and MAF against the oracle."""
import numpy as np
import pytest

import halfix
from util import oracle_liftover
from test_gpu_columns import _oracle

pytestmark = pytest.mark.gpu


def _bed(al, g, n, seed, strands="+-."):
    rng = np.random.default_rng(seed)
    seqs = [s for s in al.sequences(g) if s[2] > 0]
    lines = []
    for i in range(n):
        name, _, length = seqs[int(rng.integers(0, len(seqs)))]
        ln = int(rng.integers(1, min(length, 120) + 1))
        st = int(rng.integers(0, length - ln + 1))
        lines.append("%s\t%d\t%d\tq%d\t0\t%s\n" % (name, st, st + ln, i, strands[int(rng.integers(0, len(strands)))]))
    # whole sequences too
    for name, _, length in seqs:
        lines.append("%s\t0\t%d\tall\t0\t+\n" % (name, length))
    return "".join(lines)


@pytest.mark.parametrize("seed", list(range(8)))
def test_liftover_all_pairs(hal, oracle_bin, tmp_path, seed):
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed))
    al = hal.Alignment.open(img, device=0)
    n = al.num_genomes
    lines = 0
    for s in range(n):
        for t in range(n):
            bed = _bed(al, s, 60, seed * 100 + s * n + t)
            nd = (s + t) % 2 == 1
            got = hal.liftover_convert(al, s, bed, t, traverse_dupes=not nd)
            want = oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, no_dupes=nd)
            assert got == want, (seed, al.genome_name(s), al.genome_name(t), nd)
            lines += got.count("\n")
            if (s * n + t) % 5 == 0:
                bedp = bed.replace("\t.\n", "\t+\n")
                assert hal.liftover_convert(al, s, bedp, t, out_psl=True) == \
                    oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bedp, tmp_path, psl=True)
    assert lines > 500


@pytest.mark.parametrize("seed", [0, 3, 5])
def test_columns_all_genomes(hal, oracle_bin, tmp_path, seed):
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=7))
    al = hal.Alignment.open(img, device=0)
    for g in range(al.num_genomes):
        name = al.genome_name(g)
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        assert al.alignment_depth(g, count_dupes=True, step=3) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--countDupes", "--step", "3")
        assert al.maf_export(g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), name
        assert al.maf_export(g, unique=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--unique"), name
        assert al.maf_export(g, no_dupes=True, max_block_len=7) == \
            _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--noDupes", "--maxBlockLen", "7"), name
        sname, _, slen = al.sequences(g)[-1]
        assert al.maf_export(g, len(al.sequences(g)) - 1, start=slen // 4, length=slen // 2) == \
            _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--refSequence", sname, "--start", str(slen // 4), "--length",
                    str(slen // 2)), name


@pytest.mark.parametrize("seed,n", [(0, 22), (1, 19)])
def test_deep_caterpillar_tree(hal, oracle_bin, tmp_path, seed, n):
    """A chain of n genomes (every genome has one child): paths of up to n-1 hops.  More than 16 upward hops leave the
    chained up kernel for one launch per level; 15/16 hops sit on its limit; the way down doubles the launches."""
    img = str(tmp_path / "deep.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=n, max_children=1, root_len=600))
    al = hal.Alignment.open(img, device=0)
    ids = {al.genome_name(i): i for i in range(al.num_genomes)}
    leaf, root = ids["G%d" % (n - 1)], ids["G0"]
    pairs = [(leaf, root), (root, leaf), (leaf, ids["G2"]), (ids["G%d" % (n - 2)], ids["G%d" % (n - 2 - 16)]),
             (ids["G%d" % (n - 2)], ids["G%d" % (n - 2 - 15)]), (ids["G%d" % (n - 1)], ids["G%d" % (n - 1 - 17)]), (ids["G5"], ids["G5"])]
    lines = 0
    for s, t in pairs:
        bed = _bed(al, s, 80, seed * 7 + s + t)
        for nd in (False, True):
            got = hal.liftover_convert(al, s, bed, t, traverse_dupes=not nd)
            want = oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, no_dupes=nd)
            assert got == want, (al.genome_name(s), al.genome_name(t), nd)
            lines += got.count("\n")
    assert lines > 100
    # columns through the whole chain
    for g in (leaf, root, ids["G7"]):
        name = al.genome_name(g)
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        assert al.maf_export(g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), name
