"""The oracle (CPU restatement) against the reference's own golden vectors.  CPU only."""
import os
import subprocess

import halfix
import handbuilt_liftover as hb
from util import oracle_liftover

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _small_seed0(hal, tmp_path):
    # liftover/Makefile:62-64: halRandGen --preset small --seed 0 (storage format is irrelevant here)
    opts = hal.RandOptions.preset("small", seed=0)
    al = hal.Alignment.random(opts, device=-1)
    img = str(tmp_path / "small0.hgx")
    al.save(img)
    return al, img


def test_generator_tree_matches_reference_seed0(hal, tmp_path):
    al, _ = _small_seed0(hal, tmp_path)
    # halRandGen --preset small --seed 0 tree as written into maf/tests/expected/hal2mafSmallTest.maf's header
    assert al.newick == "((Genome_3:0)Genome_1:0,Genome_2:0)Genome_0;"


def test_generator_tree_matches_reference_seed2_config2(hal):
    # tree printed by the reference for these options (probe run of halRandGen)
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                           max_segment_length=200, min_segments=70, max_segments=140, seed=2, with_dna=True)
    al = hal.Alignment.random(opts, device=-1)
    assert al.newick == ("(((((Genome_9:0)Genome_6:2)Genome_4:0,(Genome_7:1,Genome_8:1)Genome_5:1)Genome_3:0)Genome_1:2,"
                         "Genome_2:2)Genome_0;")


def test_reference_cli_golden_bed3(hal, oracle_bin, tmp_path):
    # liftover/Makefile:46-48 halLiftoverBed3Test
    _, img = _small_seed0(hal, tmp_path)
    bed = open(os.path.join(GOLD, "ref_liftover", "test1.bed3")).read()
    want = open(os.path.join(GOLD, "ref_liftover", "halLiftoverBed3Test.bed")).read()
    assert oracle_liftover(oracle_bin, img, "Genome_0", "Genome_2", bed, tmp_path) == want


def test_reference_cli_golden_bed4_extra(hal, oracle_bin, tmp_path):
    # liftover/Makefile:59-61 halLiftoverBed4ExtraTest (--bedType 4, two pass-through columns, paralogous hits)
    _, img = _small_seed0(hal, tmp_path)
    bed = open(os.path.join(GOLD, "ref_liftover", "test1.bed4+2")).read()
    want = open(os.path.join(GOLD, "ref_liftover", "halLiftoverBed4ExtraTest.bed")).read()
    assert oracle_liftover(oracle_bin, img, "Genome_0", "Genome_2", bed, tmp_path, bed_type=4) == want


def test_reference_blockviz_golden(hal, oracle_bin, tmp_path):
    # blockViz/Makefile:52-71: blockVizTest --verbose --doSeq on halRandGen --preset small --seed 0 --minSegmentLength 3000
    # --maxSegmentLength 5000: halGetBlocksInTargetRange(Genome_2, Genome_0, Genome_0_seq, 0, 3000, HAL_QUERY_AND_TARGET_DUPS,
    # mapBackAdjacencies = 1) — the second block lies outside the range: an adjacency mapped back
    opts = hal.RandOptions.preset("small", seed=0, min_segment_length=3000, max_segment_length=5000)
    al = hal.Alignment.random(opts, device=-1)
    img = str(tmp_path / "bv.hgx")
    al.save(img)
    for name in ("blockVizMmapTests.out", "blockVizHdf5Tests.out"):
        want = open(os.path.join(GOLD, "ref_blockviz", name)).read()
        got = subprocess.run([oracle_bin, "blockviz", img, "Genome_2", "Genome_0", "Genome_0_seq", "0", "3000", "--doSeq"], check=True,
                             stdout=subprocess.PIPE).stdout.decode()
        assert got == want


def test_reference_unit_test_handbuilt(oracle_bin, tmp_path):
    # liftover/tests/halLiftoverTests.cpp:272-343 (BED6 cases): inversions, insertion, paralogy, overlap breaking
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    for src, tgt, bed, want in hb.CASES:
        assert oracle_liftover(oracle_bin, img, src, tgt, bed, tmp_path) == want, (src, tgt)


def test_oracle_asan_clean(tmp_path):
    """Reference CI runs an ASan build (.travis.yml:23-30); do the same for the restatement on a paralog-rich case."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    asan = os.path.join(root, "oracle", "_build", "hal_oracle_asan")
    for src, tgt, bed, want in hb.CASES:
        assert oracle_liftover(asan, img, src, tgt, bed, tmp_path) == want


def _oracle_maf(oracle_bin, img, tmp_path, *args):
    out = str(tmp_path / "o.maf")
    subprocess.check_call([oracle_bin, "maf", img, out] + list(args))
    return open(out).read()


def test_reference_cli_golden_hal2maf_small(hal, oracle_bin, tmp_path):
    # maf/Makefile:40-42 hal2mafSmallMMapTest: root reference, every column, DNA text, paralogous rows
    _, img = _small_seed0(hal, tmp_path)
    assert _oracle_maf(oracle_bin, img, tmp_path) == open(os.path.join(GOLD, "ref_maf", "hal2mafSmallTest.maf")).read()


def test_reference_cli_golden_hal2maf_seq_part(hal, oracle_bin, tmp_path):
    # maf/Makefile:52-54 hal2mafSeqPartTest: leaf reference, --start 1000 --length 2000
    _, img = _small_seed0(hal, tmp_path)
    got = _oracle_maf(oracle_bin, img, tmp_path, "--refGenome", "Genome_2", "--refSequence", "Genome_2_seq", "--start", "1000",
                      "--length", "2000")
    assert got == open(os.path.join(GOLD, "ref_maf", "hal2mafSeqPartTest.maf")).read()


def test_reference_cli_goldens_bed12_and_psl(hal, oracle_bin, tmp_path):
    # liftover/Makefile:38-57: halLiftoverBed12Test, Psl12Test, Psl3Test, Bed12ExtraTest
    _, img = _small_seed0(hal, tmp_path)
    d = os.path.join(GOLD, "ref_liftover")
    for inp, exp, extra in (("test1.bed12", "halLiftoverBed12Test.bed", []), ("test1.bed12+2", "halLiftoverBed12ExtraTest.bed", []),
                            ("test1.bed12", "halLiftoverPsl12Test.psl", ["--outPSL"]), ("test1.bed3", "halLiftoverPsl3Test.psl", ["--outPSL"])):
        out = str(tmp_path / "o.txt")
        subprocess.check_call([oracle_bin, "liftover", img, "Genome_0", os.path.join(d, inp), "Genome_2", out] + extra)
        assert open(out).read() == open(os.path.join(d, exp)).read(), exp


def test_reference_unit_test_bed12_psl_literals(oracle_bin, tmp_path):
    # liftover/tests/halLiftoverTests.cpp:345-373: BED12, --outPSL and --outPSLWithName literal strings
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    for src, tgt, bed, want, psl, pslname in hb.CASES12:
        inp, out = str(tmp_path / "i.bed"), str(tmp_path / "o.txt")
        open(inp, "w").write(bed)
        extra = ["--outPSLWithName"] if pslname else (["--outPSL"] if psl else [])
        subprocess.check_call([oracle_bin, "liftover", img, src, inp, tgt, out] + extra)
        assert open(out).read() == want


def test_reference_cli_golden_hal2maf_unique(hal, oracle_bin, tmp_path):
    # maf/Makefile:48-50 hal2mafSeqTest: --refGenome Genome_2 --refSequence Genome_2_seq --unique (visit cache)
    _, img = _small_seed0(hal, tmp_path)
    got = _oracle_maf(oracle_bin, img, tmp_path, "--refGenome", "Genome_2", "--refSequence", "Genome_2_seq", "--unique")
    assert got == open(os.path.join(GOLD, "ref_maf", "hal2mafSeqTest.maf")).read()


def _mp_slices(run, seq_name, seq_len, slice_size):
    """What hal2mafMP.py does with --sliceSize (maf/hal2mafMP.py:63-79 computeSlices, :81-102 concatenateSlices): one
    hal2maf --unique run per slice of the reference sequence; the first slice's file is kept whole, of the others the
    lines starting with '#' are dropped."""
    parts = []
    for i, s in enumerate(range(0, seq_len, slice_size)):
        t = run(s, min(slice_size, seq_len - s))
        if i > 0:
            t = "".join(x for x in t.splitlines(True) if not x.startswith("#"))
        parts.append(t)
    return "".join(parts)


def test_reference_goldens_of_the_multiprocess_driver(hal, oracle_bin, tmp_path):
    """The expected files of hal2mafMP.py's tests (maf/Makefile:62-72; the driver is deprecated, its goldens are still in
    the tree): --targetGenomes, --refTargets, and a reference sequence cut into 250-base slices that are exported
    separately and concatenated — the reference's own way of sharding the column path."""
    al, img = _small_seed0(hal, tmp_path)
    d = os.path.join(GOLD, "ref_maf")
    g3 = al.genome_id("Genome_3")
    name, _, ln = al.sequences(g3)[0]
    got = _oracle_maf(oracle_bin, img, tmp_path, "--refGenome", "Genome_3", "--targetGenomes", "Genome_1,Genome_2", "--refSequence", name,
                      "--start", "0", "--length", str(ln), "--unique")
    assert got == open(os.path.join(d, "hal2mafMPTargetGenomesTest.maf")).read()
    got = _oracle_maf(oracle_bin, img, tmp_path, "--refTargets", os.path.join(d, "small-Genome_0.bed"), "--unique")
    assert got == open(os.path.join(d, "hal2mafMPRefTargetsGenomesTest.maf")).read()
    g0 = al.genome_id("Genome_0")
    name, _, ln = al.sequences(g0)[0]
    got = _mp_slices(lambda s, l: _oracle_maf(oracle_bin, img, tmp_path, "--refGenome", "Genome_0", "--refSequence", name, "--start", str(s),
                                              "--length", str(l), "--unique"), name, ln, 250)
    assert got == open(os.path.join(d, "hal2mafMPBySeqTest_Genome_0_seq.maf")).read()


def test_reference_unit_tests_of_the_column_iterator(oracle_bin, tmp_path):
    """api/tests/halColumnIteratorTest.cpp: the Depth, Dup and Inv hand-built alignments and what the reference asserts about
    every column of them (tests/golden/handbuilt_columns.py)."""
    import handbuilt_columns as hc
    for name, build, check, refs in hc.CASES:
        img = str(tmp_path / (name + ".hgx"))
        halfix.write_hgx(img, build())
        for ref in refs:
            out = subprocess.run([oracle_bin, "columns", img, ref], check=True, stdout=subprocess.PIPE).stdout.decode()
            lines = out.splitlines()
            assert len(lines) == 100
            for line in lines:
                f = line.split()
                rows = [(x.split(":")[0], int(x.split(":")[1]), x.split(":")[2] == "-") for x in f[1:]]
                check(ref, int(f[0]), rows)


def test_reference_unit_tests_of_the_column_iterator_with_gaps(oracle_bin, tmp_path):
    """api/tests/halColumnIteratorTest.cpp:459-933: Gap, MultiGap and MultiGapInv — the column iterator with maxInsertLength 1000
    walks the deleted ranges of the ancestors as columns of their own between two reference columns (the indel stack)."""
    import handbuilt_columns as hc
    for name, build, check, ref, ncol in hc.GAP_CASES:
        img = str(tmp_path / (name + ".hgx"))
        halfix.write_hgx(img, build())
        out = subprocess.run([oracle_bin, "columns", img, ref, "--maxRefGap", "1000"], check=True, stdout=subprocess.PIPE).stdout.decode()
        lines = out.splitlines()
        assert len(lines) == ncol, (name, len(lines))
        for line in lines:
            f = line.split()
            check(int(f[0]), [(x.split(":")[0], int(x.split(":")[1]), x.split(":")[2] == "-") for x in f[1:]])
        # without the stack the iterator gives the reference's own columns only
        plain = subprocess.run([oracle_bin, "columns", img, ref], check=True, stdout=subprocess.PIPE).stdout.decode().splitlines()
        assert len(plain) == 8


def test_reference_unit_test_extra_paralogs_coalescence_limit(oracle_bin, tmp_path):
    """api/tests/halMappedSegmentTest.cpp:478-611: the known answer for a coalescence limit above the MRCA."""
    img = str(tmp_path / "xp.hgx")
    halfix.write_hgx(img, hb.extra_paralogs_genomes())
    run = lambda *a: subprocess.run([oracle_bin, "blocks", img, "grandChild2", "grandChild1", "0", "2"] + list(a), check=True,
                                    stdout=subprocess.PIPE).stdout.decode()
    assert run() == hb.EXTRA_PARALOGS_DEFAULT
    assert run("--coalescenceLimit", "root") == hb.EXTRA_PARALOGS_ROOT_LIMIT
    assert run("--coalescenceLimit", "parent") == hb.EXTRA_PARALOGS_DEFAULT
