"""hal2maf --unique on the device path (hgx_column_kernels.hpp: k_column_unique_count): which columns ColumnIterator's visit cache
lets the iterator walk, and which of them hal2maf writes (api/impl/halColumnIterator.cpp:208-212, 749-819;
maf/impl/halMafExport.cpp:52-64), is decided per column on the device, and the written columns go through the run-compressed
export.  Held against the oracle's faithful visit cache and against the library's own replay of that cache on the host
(HGX_UNIQUE_REPLAY=1), on whole sequences, on sub-ranges that begin inside a genome (reference bases left of the range: the
columns that are walked but not written), with the options that cut the walk, and across the ends of device batches."""
import os

import pytest

import halfix
from test_gpu_columns import _oracle, _rand

pytestmark = pytest.mark.gpu


def _env(**kw):
    class E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            os.environ.update({k: str(v) for k, v in kw.items()})

        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return E()


VARIANTS = [(dict(), []), (dict(no_dupes=True), ["--noDupes"]), (dict(only_orthologs=True), ["--onlyOrthologs"]),
            (dict(no_ancestors=True), ["--noAncestors"]), (dict(max_block_len=7), ["--maxBlockLen", "7"]),
            (dict(keep_empty_ref_blocks=True), ["--keepEmptyRefBlocks"])]


@pytest.mark.parametrize("seed", list(range(6)))
def test_unique_multiseq_every_reference(hal, oracle_bin, tmp_path, seed):
    """several sequences per genome: a sequence's range begins inside the genome, so paralogs in the sequences before it make
    columns that are walked and not written"""
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=6))
    al = hal.Alignment.open(img, device=0)
    n_bytes = 0
    for g in range(al.num_genomes):
        nm = al.genome_name(g)
        leaf = not al.genome_children(g)
        for kw, args in VARIANTS:
            if kw.get("no_ancestors") and not leaf:
                continue
            want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique", *args)
            got = al.maf_export(g, unique=True, **kw)
            assert got == want, (nm, kw)
            n_bytes += len(got)
        with _env(HGX_UNIQUE_REPLAY=1):
            assert al.maf_export(g, unique=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique"), nm
        with _env(HGX_MAF_CHUNK=13):  # (device batches of 13 columns: runs, skipped stretches and blocks cross their ends)
            assert al.maf_export(g, unique=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique"), nm
        # sub-ranges of every sequence
        for si, (sname, _, slen) in enumerate(al.sequences(g)):
            if slen < 4:
                continue
            for a, ln in ((slen // 3, slen // 2), (slen - 2, 2), (1, slen - 1)):
                want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--refSequence", sname, "--start", str(a), "--length",
                               str(ln), "--unique")
                assert al.maf_export(g, si, start=a, length=ln, unique=True) == want, (nm, sname, a, ln)
    assert n_bytes > 0


@pytest.mark.parametrize("seed", [1, 4, 7])
def test_unique_halrandgen_with_targets_and_slices(hal, oracle_bin, tmp_path, seed):
    al, img = _rand(hal, tmp_path, seed, dna=True)
    for g in range(al.num_genomes):
        nm = al.genome_name(g)
        n = al.genome_length(g)
        sname = al.sequences(g)[0][0]
        assert al.maf_export(g, unique=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique"), nm
        tg = [t for t in (0, al.num_genomes - 1, al.num_genomes // 2) if t != g]
        tnames = ",".join(al.genome_name(t) for t in tg)
        assert al.maf_export(g, unique=True, targets=tg) == \
            _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique", "--targetGenomes", tnames), (nm, tnames)
        # hal2mafMP's slices (maf/hal2mafMP.py:63-79): every slice an export of its own
        step = max(1, n // 5)
        for a in range(0, n, step):
            ln = min(step, n - a)
            want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--refSequence", sname, "--start", str(a), "--length",
                           str(ln), "--unique", "--noDupes" if a % 2 else "--onlySequenceNames")
            got = al.maf_export(g, 0, start=a, length=ln, unique=True, **({"no_dupes": True} if a % 2 else {"only_sequence_names": True}))
            assert got == want, (nm, a, ln)
            with _env(HGX_UNIQUE_REPLAY=1):
                assert al.maf_export(g, 0, start=a, length=ln, unique=True,
                                     **({"no_dupes": True} if a % 2 else {"only_sequence_names": True})) == want, (nm, a, ln)


def test_unique_fast_path_is_the_one_that_runs(hal, tmp_path):
    """the run-compressed path reports its columns through the library's statistics: a --unique export of 200 k columns takes
    well under the replay's time (no per-column rows cross PCIe)"""
    import time
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                           max_segment_length=200, min_segments=3000, max_segments=6000, seed=2, with_dna="fast")
    al = hal.Alignment.random(opts, device=0)
    g = al.genome_id("Genome_9")
    n = min(200000, al.genome_length(g))
    al.maf_export_bytes(g, 0, start=0, length=1000, no_ancestors=True)
    fast = al.maf_export(g, 0, start=0, length=n, unique=True, no_ancestors=True)
    t0 = time.perf_counter()
    fast = al.maf_export(g, 0, start=0, length=n, unique=True, no_ancestors=True)
    t_fast = time.perf_counter() - t0
    with _env(HGX_UNIQUE_REPLAY=1):
        t0 = time.perf_counter()
        slow = al.maf_export(g, 0, start=0, length=n, unique=True, no_ancestors=True)
        t_slow = time.perf_counter() - t0
    assert fast == slow
    print("unique: device path %.3f s, host replay %.3f s for %d columns" % (t_fast, t_slow, n))
