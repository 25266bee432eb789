"""hal2maf --maxRefGap (ColumnIterator with maxInsertLength > 0: the stack of inserted and deleted ranges walked between two
reference columns, api/impl/halColumnIterator.cpp:65-144, 357-405, with Rearrangement's deletion and insertion cycles,
api/impl/halRearrangement.cpp:133-176, 386-516) through the HIP path — the device walks the columns and reports the indels
their walks meet (hgx_gap_kernels.hpp), the host replays the iterator's stack and visit cache — against the known answers of the
reference's unit tests and against the oracle."""
import os
import subprocess

import pytest

import halfix
import handbuilt_columns as hc
from test_gpu_columns import _oracle, _rand

pytestmark = pytest.mark.gpu


# what the iteration the unit tests check looks like as MAF (one block: the deleted bases are columns between dad's)
KNOWN = {"gap": "a\ns\tdad.dseq\t0\t8\t+\t8\tACGT----GGGG\ns\tgrandpa.gseq\t0\t12\t+\t12\tACGTAAAAGGGG\n\n",
         "multi_gap": "a\ns\tdad.dseq\t0\t8\t+\t8\tACGT--------GGGG\ns\tadam.aseq\t0\t16\t+\t16\tACGTAAAATTTTGGGG\n"
                      "s\tgrandpa.gseq\t0\t12\t+\t12\tACGTAAAA----GGGG\n\n"}


def test_reference_unit_tests_gap_multigap_multigapinv(hal, oracle_bin, tmp_path):
    """api/tests/halColumnIteratorTest.cpp:459-933 (tests/golden/handbuilt_columns.py: GAP_CASES, checked column by column on the
    oracle in test_oracle_golden.py): iterated from dad with maxInsertLength 1000 the deleted bases of the ancestors come between
    dad's columns.  The MAF of the HIP path is the oracle's byte for byte, also with the blocks whose reference row is all gaps."""
    for name, build, check, ref, ncol in hc.GAP_CASES:
        img = str(tmp_path / (name + ".hgx"))
        halfix.write_hgx(img, build())
        al = hal.Alignment.open(img, device=0)
        g = al.genome_id(ref)
        got = al.maf_export(g, max_ref_gap=1000)
        assert got == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", ref, "--maxRefGap", "1000"), name
        if name in KNOWN:
            assert got.endswith(KNOWN[name]) and got.count("a\n") == 1
        kept = al.maf_export(g, max_ref_gap=1000, keep_empty_ref_blocks=True)
        top = "grandpa" if name == "gap" else "adam"
        assert sum(int(l.split("\t")[3]) for l in kept.splitlines() if l.startswith("s\t" + top)) == ncol  # every ancestor base once
        assert al.maf_export(g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", ref)


@pytest.mark.parametrize("seed", list(range(6)))
def test_multiseq_alignments_against_the_oracle(hal, oracle_bin, tmp_path, seed):
    """insertions, deletions, inversions and duplications of the independent generator: every genome as the reference, small and
    large gaps, with the options that filter rows (the visit cache looks at the filtered bases too)"""
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=6))
    al = hal.Alignment.open(img, device=0)
    n_bytes = 0
    for g in range(al.num_genomes):
        nm = al.genome_name(g)
        leaf = not al.genome_children(g)
        for gap in (3, 25, 1000):
            variants = [dict(), dict(no_dupes=True), dict(unique=True), dict(only_orthologs=True)]
            if leaf:
                variants.append(dict(no_ancestors=True))
            for kw in variants[: 2 + (seed + g + gap) % 4]:
                args = ["--refGenome", nm, "--maxRefGap", str(gap)]
                args += ["--noDupes"] if kw.get("no_dupes") else []
                args += ["--unique"] if kw.get("unique") else []
                args += ["--onlyOrthologs"] if kw.get("only_orthologs") else []
                args += ["--noAncestors"] if kw.get("no_ancestors") else []
                got = al.maf_export(g, max_ref_gap=gap, **kw)
                assert got == _oracle(oracle_bin, "maf", img, tmp_path, *args), (seed, nm, gap, kw)
                n_bytes += len(got)
    assert n_bytes > 20000


@pytest.mark.parametrize("seed", [2, 5])
def test_randgen_alignments_against_the_oracle(hal, oracle_bin, tmp_path, seed):
    al, img = _rand(hal, tmp_path, seed)
    for g in range(al.num_genomes):
        nm = al.genome_name(g)
        for gap in (10, 200):
            assert al.maf_export(g, max_ref_gap=gap) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--maxRefGap", str(gap)), (nm, gap)
    # a target set (scope and row filter), a sub-range, sequence names only
    leafs = [g for g in range(al.num_genomes) if not al.genome_children(g)]
    ref, tg = leafs[0], leafs[-1]
    name, _, length = al.sequences(ref)[0]
    got = al.maf_export(ref, ref_sequence=0, start=length // 4, length=length // 2, targets=[tg], max_ref_gap=50, only_sequence_names=True,
                        no_ancestors=True)
    want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", al.genome_name(ref), "--refSequence", name, "--start", str(length // 4),
                   "--length", str(length // 2), "--targetGenomes", al.genome_name(tg), "--maxRefGap", "50", "--onlySequenceNames", "--noAncestors")
    assert got == want


def test_real_data_and_cli(hal, oracle_bin, tmp_path):
    """evolver mouse / rat (real indels) and the hal2maf twin's --maxRefGap"""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    try:
        al = hal.Alignment.open(os.path.join(gold, "ref_hdf5", "mr.hal"), device=0)
    except hal.HgxError as e:
        if "HDF5 C library" in str(e):
            pytest.skip("libhdf5 not loadable here")
        raise
    img = str(tmp_path / "mr.hgx")
    al.save(img)
    g = al.genome_id("simMouse_chr6")
    name, _, length = al.sequences(g)[0]
    n = min(length, 60000)
    want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", "simMouse_chr6", "--refSequence", name, "--start", "0", "--length", str(n),
                   "--maxRefGap", "100", "--noAncestors")
    assert al.maf_export(g, ref_sequence=0, start=0, length=n, max_ref_gap=100, no_ancestors=True) == want
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hal_amd", "_build", "hal2maf")
    out = str(tmp_path / "cli.maf")
    subprocess.check_call([tool, "--refGenome", "simMouse_chr6", "--refSequence", name, "--start", "0", "--length", str(n), "--maxRefGap", "100",
                           "--noAncestors", img, out])
    assert open(out).read() == want


def test_global_export_against_the_oracle(hal, oracle_bin, tmp_path):
    """hal2maf --global (MafExport::convertEntireAlignment, maf/impl/halMafExport.cpp:90-153): the leaves one after the other, each
    with the visit cache of the ones before; the replay over the device's unfiltered columns gives the oracle's bytes (the reference
    holds no expected file for this option: oracle only)."""
    import subprocess as sp
    al, img = _rand(hal, tmp_path, 2)
    assert al.maf_export_global() == _oracle(oracle_bin, "maf", img, tmp_path, "--global")
    assert al.maf_export_global(no_ancestors=True, no_dupes=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--global", "--noAncestors", "--noDupes")
    for seed in (0, 4):
        img2 = str(tmp_path / ("g%d.hgx" % seed))
        halfix.write_hgx(img2, halfix.random_multiseq_alignment(seed, n_genomes=7))
        al2 = hal.Alignment.open(img2, device=0)
        assert al2.maf_export_global() == _oracle(oracle_bin, "maf", img2, tmp_path, "--global"), seed
        assert al2.maf_export_global(no_ancestors=True, only_sequence_names=True, max_block_len=20) == \
            _oracle(oracle_bin, "maf", img2, tmp_path, "--global", "--noAncestors", "--onlySequenceNames", "--maxBlockLen", "20"), seed
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hal_amd", "_build", "hal2maf")
    out = str(tmp_path / "cli_global.maf")
    sp.check_call([tool, "--global", "--noAncestors", img, out])
    assert open(out).read() == _oracle(oracle_bin, "maf", img, tmp_path, "--global", "--noAncestors")
