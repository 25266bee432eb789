"""BlockMapper seam (liftover/inc/halBlockMapper.h:30-40, init + map + getMap without adjacencies) through hgx_block_map:
the members of the mapped set, in set order, HIP path vs the oracle's restatement of BlockMapper::map."""
import subprocess

import numpy as np
import pytest

import halfix
import handbuilt_liftover as hb
from test_gpu_liftover import _rand_alignment

pytestmark = pytest.mark.gpu


def _oracle_blocks(oracle_bin, img, ref, query, first, last, *flags):
    return subprocess.run([oracle_bin, "blocks", img, ref, query, str(first), str(last)] + list(flags), check=True,
                          stdout=subprocess.PIPE).stdout.decode()


def _text(al, query, recs):
    seqs = al.sequences(query)
    return "".join("%s\t%d\t%d\t%d\t%s\t%s\n" % (seqs[int(r["tgt_seq"])][0], r["tgt_start"], r["tgt_end"], r["src_start"],
                                                 r["strand"].decode(), "-" if r["tgt_reversed"] else "+") for r in recs)


def _check(hal, oracle_bin, al, img, ref, query, first, last, **kw):
    flags = []
    if kw.get("target_reversed"):
        flags.append("--reversed")
    if kw.get("do_dupes") is False:
        flags.append("--noDupes")
    if kw.get("min_length"):
        flags += ["--minLength", str(kw["min_length"])]
    if kw.get("coalescence_limit", -1) >= 0:
        flags += ["--coalescenceLimit", al.genome_name(kw["coalescence_limit"])]
    got = _text(al, query, al.block_map(ref, query, first, last, **kw))
    want = _oracle_blocks(oracle_bin, img, al.genome_name(ref), al.genome_name(query), first, last, *flags)
    assert got == want, (al.genome_name(ref), al.genome_name(query), first, last, kw)
    return got.count("\n")


def test_handbuilt_all_pairs(hal, oracle_bin, tmp_path):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    n = al.num_genomes
    total = 0
    for r in range(n):
        length = al.genome_length(r)
        for q in range(n):
            for first, last in ((0, length - 1), (3, min(length - 1, 41)), (length // 2, length // 2)):
                for rev in (False, True):
                    total += _check(hal, oracle_bin, al, img, r, q, first, last, target_reversed=rev)
            total += _check(hal, oracle_bin, al, img, r, q, 0, length - 1, do_dupes=False)
            total += _check(hal, oracle_bin, al, img, r, q, 0, length - 1, min_length=8)
    assert total > 200


@pytest.mark.parametrize("seed", [2, 6])
def test_randgen_ranges(hal, oracle_bin, tmp_path, seed):
    al, img = _rand_alignment(hal, tmp_path, seed)
    n = al.num_genomes
    rng = np.random.default_rng(seed)
    total = 0
    for _ in range(40):
        r, q = int(rng.integers(0, n)), int(rng.integers(0, n))
        length = al.genome_length(r)
        if length == 0:
            continue
        ln = int(rng.integers(1, min(length, 4000) + 1))
        first = int(rng.integers(0, length - ln + 1))
        kw = {"target_reversed": bool(rng.integers(0, 2)), "do_dupes": bool(rng.integers(0, 4))}
        m = al.mrca(r, q)
        if rng.integers(0, 3) == 0 and al.genome_parent(m) >= 0:
            kw["coalescence_limit"] = al.genome_parent(m)
        total += _check(hal, oracle_bin, al, img, r, q, first, first + ln - 1, **kw)
    assert total > 300


def test_range_errors(hal, tmp_path):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    with pytest.raises(hal.HgxError, match="out of bounds"):
        al.block_map(1, 2, 5, al.genome_length(1))
    with pytest.raises(hal.HgxError, match="out of bounds"):
        al.block_map(1, 2, 9, 3)
