"""halLiftover --coalescenceLimit (mapRecursiveParalogies, api/impl/halSegmentMapper.cpp:525-576): paralogs that coalesce
above the MRCA, up to the given ancestor, are included.  HIP path vs oracle on halRandGen alignments with inversions
(--maxBranchLength 3), on the independent multi-sequence generator (real duplications at every level), and on the
hand-built alignment of the reference's unit test."""
import numpy as np
import pytest

import halfix
import handbuilt_liftover as hb
from util import oracle_liftover, random_bed
from test_gpu_liftover import _rand_alignment
from test_gpu_multiseq import _bed

pytestmark = pytest.mark.gpu


def _ancestors(al, g):
    out = []
    g = al.genome_parent(g)
    while g >= 0:
        out.append(g)
        g = al.genome_parent(g)
    return out


def _check(hal, oracle_bin, al, img, s, t, limit, bed, tmp_path, **kw):
    got = hal.liftover_convert(al, s, bed, t, coalescence_limit=limit, **kw)
    okw = {}
    if kw.get("traverse_dupes") is False:
        okw["no_dupes"] = True
    if kw.get("out_psl"):
        okw["psl"] = True
    want = oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path,
                           coalescence_limit=al.genome_name(limit), **okw)
    assert got == want, (al.genome_name(s), al.genome_name(t), al.genome_name(limit), kw)
    return got


@pytest.mark.parametrize("seed", [2, 5, 6])
def test_randgen_every_limit(hal, oracle_bin, tmp_path, seed):
    al, img = _rand_alignment(hal, tmp_path, seed)
    n = al.num_genomes
    rng = np.random.default_rng(seed)
    checked = extra = 0
    for _ in range(14):
        s, t = int(rng.integers(0, n)), int(rng.integers(0, n))
        name, _, length = al.sequences(s)[0]
        if length == 0:
            continue
        m = al.mrca(s, t)
        bed = random_bed(name, length, 150, 1, 300, seed * 31 + checked)
        base = hal.liftover_convert(al, s, bed, t)
        for limit in [m] + _ancestors(al, m):
            got = _check(hal, oracle_bin, al, img, s, t, limit, bed, tmp_path)
            extra += got != base
            checked += 1
        # with --noDupes the limit changes nothing (mapSource, halSegmentMapper.cpp:616-621)
        for limit in _ancestors(al, m)[:1]:
            _check(hal, oracle_bin, al, img, s, t, limit, bed, tmp_path, traverse_dupes=False)
    assert checked > 10 and extra > 0  # some limit really added paralogs


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_multiseq_every_limit(hal, oracle_bin, tmp_path, seed):
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=7))
    al = hal.Alignment.open(img, device=0)
    n = al.num_genomes
    checked = extra = 0
    for s in range(n):
        for t in range(n):
            m = al.mrca(s, t)
            anc = _ancestors(al, m)
            if not anc:
                continue
            bed = _bed(al, s, 40, seed * 100 + s * n + t)
            base = hal.liftover_convert(al, s, bed, t)
            for limit in anc:
                got = _check(hal, oracle_bin, al, img, s, t, limit, bed, tmp_path)
                extra += got != base
                checked += 1
            if (s + t) % 3 == 0:
                _check(hal, oracle_bin, al, img, s, t, anc[-1], bed.replace("\t.\n", "\t+\n"), tmp_path, out_psl=True)
    assert checked > 8 and extra >= 0


def test_handbuilt_and_errors(hal, oracle_bin, tmp_path):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    ids = {al.genome_name(i): i for i in range(al.num_genomes)}
    bed = "Sequence\t0\t70\tall\t0\t+\nSequence\t5\t33\tpart\t0\t-\n"
    for s, t in (("leaf2", "leaf3"), ("leaf3", "leaf2"), ("leaf2", "child1"), ("child1", "leaf2"), ("leaf2", "leaf2")):
        _check(hal, oracle_bin, al, img, ids[s], ids[t], ids["root"], bed, tmp_path)
    # a limit that is not an ancestor of the MRCA: the reference runs into the root while climbing (halSegmentMapper.cpp:541)
    with pytest.raises(hal.HgxError, match="Hit root genome"):
        hal.liftover_convert(al, ids["leaf2"], bed, ids["leaf1"], coalescence_limit=ids["child1"])
