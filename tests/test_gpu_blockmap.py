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


def test_reference_unit_test_extra_paralogs_coalescence_limit(hal, tmp_path):
    """api/tests/halMappedSegmentTest.cpp:478-611 through hgx_block_map: one homology by default, all three paralogs with the
    root as coalescence limit."""
    img = str(tmp_path / "xp.hgx")
    halfix.write_hgx(img, hb.extra_paralogs_genomes())
    al = hal.Alignment.open(img, device=0)
    ids = {al.genome_name(i): i for i in range(al.num_genomes)}
    r, q = ids["grandChild2"], ids["grandChild1"]
    assert _text(al, q, al.block_map(r, q, 0, 2)) == hb.EXTRA_PARALOGS_DEFAULT
    assert _text(al, q, al.block_map(r, q, 0, 2, coalescence_limit=ids["root"])) == hb.EXTRA_PARALOGS_ROOT_LIMIT
    assert _text(al, q, al.block_map(r, q, 0, 2, coalescence_limit=ids["parent"])) == hb.EXTRA_PARALOGS_DEFAULT


def _pairs_from_blocks(al, ref, tgt):
    """(reference position, target position) of every aligned base pair according to the block mapper"""
    n = al.genome_length(ref)
    recs = al.block_map(ref, tgt, 0, n - 1)
    tstart = {i: s[1] for i, s in enumerate(al.sequences(tgt))}
    out = []
    for r in recs:
        ln = int(r["tgt_end"] - r["tgt_start"])
        ss = tstart[int(r["tgt_seq"])]
        src = np.arange(ln, dtype=np.int64) + int(r["src_start"])
        assert r["strand"] == b"+"
        tg = (int(r["tgt_end"]) - 1 - np.arange(ln, dtype=np.int64) if r["tgt_reversed"] else int(r["tgt_start"]) + np.arange(ln, dtype=np.int64)) + ss
        out.append(np.stack([src, tg], axis=1))
    return np.unique(np.concatenate(out), axis=0) if out else np.zeros((0, 2), np.int64)


def _pairs_from_columns(al, ref, tgt):
    """the same according to the column engine: the target bases in the column of every reference position"""
    n = al.genome_length(ref)
    off, rows = al.column_rows(ref, 0, n, targets=[tgt])
    col = np.repeat(np.arange(n, dtype=np.int64), np.diff(off).astype(np.int64))
    keep = rows["genome"] == tgt
    if ref == tgt:  # the reference base itself is in its own column; the block mapper reports it as well
        pass
    return np.unique(np.stack([col[keep], rows["pos"][keep].astype(np.int64)], axis=1), axis=0)


@pytest.mark.parametrize("seed", [1, 4])
def test_columns_and_blocks_agree(hal, tmp_path, seed):
    """MappedSegmentColCompareTest (api/tests/halMappedSegmentTest.cpp:615-760) as a property of the two HIP engines: for
    every pair of genomes, the homologies a reference base has in the target are the same whether they come from the
    column walk (ColumnIterator) or from the block mapper (halMapSegment); orientation is not compared (:744-755)."""
    al, _ = _rand_alignment(hal, tmp_path, seed, max_genomes=8, min_segs=60, max_segs=200)
    n = al.num_genomes
    checked = 0
    for r in range(n):
        if al.genome_length(r) == 0:
            continue
        for q in range(n):
            if al.genome_length(q) == 0 or r == q:
                continue
            a, b = _pairs_from_columns(al, r, q), _pairs_from_blocks(al, r, q)
            assert a.shape == b.shape and (a == b).all(), (al.genome_name(r), al.genome_name(q), a.shape, b.shape)
            checked += len(a)
    assert checked > 10000
