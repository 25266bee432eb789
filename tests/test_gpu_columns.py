"""Parity of the HIP column engine (halAlignmentDepth, hal2maf) with the oracle and the reference's goldens."""
import os
import subprocess

import numpy as np
import pytest

import halfix
import handbuilt_liftover as hb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _oracle(oracle_bin, cmd, img, tmp_path, *args):
    out = str(tmp_path / ("o." + cmd))
    if cmd == "maf":
        subprocess.check_call([oracle_bin, "maf", img, out] + list(args))
    else:
        subprocess.check_call([oracle_bin, "depth", img, args[0], out] + list(args[1:]))
    return open(out).read()


def _rand(hal, tmp_path, seed, dna=True, **kw):
    o = dict(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10, max_segment_length=60,
             min_segments=60, max_segments=160, seed=seed, with_dna=dna)
    o.update(kw)
    al = hal.Alignment.random(hal.RandOptions(**o), device=0)
    img = str(tmp_path / ("c%d.hgx" % seed))
    al.save(img)
    return al, img


def test_reference_cli_goldens_hal2maf(hal, tmp_path):
    al = hal.Alignment.random(hal.RandOptions.preset("small", seed=0), device=0)
    assert al.maf_export(al.genome_id("Genome_0")) == open(os.path.join(GOLD, "ref_maf", "hal2mafSmallTest.maf")).read()
    g2 = al.genome_id("Genome_2")
    assert al.maf_export(g2, 0, start=1000, length=2000) == open(os.path.join(GOLD, "ref_maf", "hal2mafSeqPartTest.maf")).read()
    # maf/Makefile:48-50 hal2mafSeqTest: --unique
    assert al.maf_export(g2, 0, unique=True) == open(os.path.join(GOLD, "ref_maf", "hal2mafSeqTest.maf")).read()


@pytest.mark.parametrize("seed", [2, 5, 6])
def test_depth_all_genomes_vs_oracle(hal, oracle_bin, tmp_path, seed):
    al, img = _rand(hal, tmp_path, seed, dna=False)
    for g in range(al.num_genomes):
        name = al.genome_name(g)
        if al.genome_length(g) == 0:
            continue
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        assert al.alignment_depth(g, count_dupes=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--countDupes"), name
        if not al.genome_children(g):
            assert al.alignment_depth(g, no_ancestors=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--noAncestors")


def test_depth_options_vs_oracle(hal, oracle_bin, tmp_path):
    al, img = _rand(hal, tmp_path, 2, dna=False)
    g9, g8, g2, g5 = (al.genome_id(n) for n in ("Genome_9", "Genome_8", "Genome_2", "Genome_5"))
    assert al.alignment_depth(g9, 0, start=100, length=777) == \
        _oracle(oracle_bin, "depth", img, tmp_path, "Genome_9", "--refSequence", "Genome_9_seq", "--start", "100", "--length", "777")
    assert al.alignment_depth(g9, step=7) == _oracle(oracle_bin, "depth", img, tmp_path, "Genome_9", "--step", "7")
    assert al.alignment_depth(g9, 0, start=5, length=70, step=10) == \
        _oracle(oracle_bin, "depth", img, tmp_path, "Genome_9", "--refSequence", "Genome_9_seq", "--start", "5", "--length", "70",
                "--step", "10")
    assert al.alignment_depth(g9, targets=[g8, g2]) == \
        _oracle(oracle_bin, "depth", img, tmp_path, "Genome_9", "--targetGenomes", "Genome_8,Genome_2")
    assert al.alignment_depth(g5, targets=[g8]) == _oracle(oracle_bin, "depth", img, tmp_path, "Genome_5", "--targetGenomes", "Genome_8")


def test_depth_handbuilt_and_columns_api(hal, oracle_bin, tmp_path):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    for g in range(al.num_genomes):
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, al.genome_name(g))
    # raw per-column API agrees with the wig text
    g = al.genome_id("leaf2")
    vals = al.columns_depth(g, 0, 70)
    assert "\n".join(map(str, vals)) + "\n" == al.alignment_depth(g).split("\n", 1)[1]
    off, rows = al.column_rows(g, 0, 70)
    assert len(off) == 71 and off[-1] == len(rows)
    assert np.array_equal(np.diff(off).astype(np.int32) - 1, al.columns_depth(g, 0, 70, count_dupes=True))
    # the first row of every column is the reference base itself
    assert np.all(rows["genome"][off[:-1]] == g) and np.array_equal(rows["pos"][off[:-1]], np.arange(70))


@pytest.mark.parametrize("seed", [2, 5])
def test_maf_vs_oracle(hal, oracle_bin, tmp_path, seed):
    al, img = _rand(hal, tmp_path, seed, dna=True)
    for name in ("Genome_0", "Genome_1", "Genome_2", al.genome_name(al.num_genomes - 1)):
        g = al.genome_id(name)
        assert al.maf_export(g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), name
    leaf = al.genome_name(al.num_genomes - 1)
    g = al.genome_id(leaf)
    assert al.maf_export(g, no_ancestors=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", leaf, "--noAncestors")
    assert al.maf_export(g, no_dupes=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", leaf, "--noDupes")
    assert al.maf_export(g, only_orthologs=True, only_sequence_names=True, max_block_len=50) == \
        _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", leaf, "--onlyOrthologs", "--onlySequenceNames", "--maxBlockLen", "50")
    assert al.maf_export(g, targets=[0, 2]) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", leaf, "--targetGenomes",
                                                        "Genome_0,Genome_2")
    # --unique on whole sequences and on sub-ranges (what hal2mafMP's slices run, maf/hal2mafMP.py:63-79)
    for name in ("Genome_0", "Genome_1", leaf):
        gg = al.genome_id(name)
        assert al.maf_export(gg, unique=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--unique"), name
        n = al.genome_length(gg)
        sname = al.sequences(gg)[0][0]
        for a, ln in ((0, n // 3), (n // 3, n // 3), (2 * (n // 3), n - 2 * (n // 3))):
            assert al.maf_export(gg, 0, start=a, length=ln, unique=True) == \
                _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--refSequence", sname, "--start", str(a), "--length",
                        str(ln), "--unique"), (name, a, ln)


def test_maf_handbuilt_inversions(hal, oracle_bin, tmp_path):
    img = str(tmp_path / "hand.hgx")
    halfix.write_hgx(img, hb.genomes())
    al = hal.Alignment.open(img, device=0)
    for g in range(al.num_genomes):
        assert al.maf_export(g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", al.genome_name(g)), al.genome_name(g)


def test_depth_properties_at_scale(hal):
    """2 M columns on a 10-genome alignment: depth is bounded by the genome count, countDupes >= unique count,
    restricting targets never raises the count, and results are deterministic."""
    o = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                        max_segment_length=200, min_segments=20000, max_segments=40000, seed=2, with_dna=False)
    al = hal.Alignment.random(o, device=0)
    g = al.genome_id("Genome_9")
    n = min(2000000, al.genome_length(g))
    d = al.columns_depth(g, 0, n)
    dd = al.columns_depth(g, 0, n, count_dupes=True)
    dt = al.columns_depth(g, 0, n, targets=[al.genome_id("Genome_2"), al.genome_id("Genome_8")])
    assert d.min() >= 0 and d.max() <= al.num_genomes - 1 and np.all(dd >= d) and np.all(dt <= d)
    assert np.array_equal(d, al.columns_depth(g, 0, n)) and d.max() > 3
    assert np.array_equal(al.columns_depth(g, 0, n, step=1)[::5][: n // 5], al.columns_depth(g, 0, n // 5, step=5))


def test_maf_ref_targets_vs_oracle(hal, oracle_bin, tmp_path):
    """hal2maf --refTargets (maf/impl/halMafBed.cpp): BED3 intervals and BED12 blocks of the reference, one shared MafExport."""
    al, img = _rand(hal, tmp_path, 5, dna=True)
    leaf = al.genome_name(al.num_genomes - 1)
    g = al.genome_id(leaf)
    sname, _, n = al.sequences(g)[0]
    bed = ("%s\t10\t300\n" % sname + "%s\t250\t900\tx\t0\t-\n" % sname + "nosuch\t1\t5\n" + "%s\t5\t%d\n" % (sname, n + 5) +
           "%s\t1000\t1600\tb\t0\t+\t1000\t1600\t0\t3\t50,70,0,\t0,200,400,\n" % sname)
    bedfile = tmp_path / "t.bed"
    bedfile.write_text(bed)
    assert al.maf_export(g, ref_targets_bed=bed) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", leaf, "--refTargets", str(bedfile))
    assert al.maf_export(g, ref_targets_bed=bed, unique=True, no_ancestors=True) == \
        _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", leaf, "--refTargets", str(bedfile), "--unique", "--noAncestors")


def test_reference_goldens_of_the_multiprocess_driver(hal, tmp_path):
    """Product path against the expected files of the reference's sliced runner (hal2mafMP.py, maf/Makefile:62-72): target
    genomes, --refTargets, and 250-base slices exported separately and concatenated."""
    from test_oracle_golden import _mp_slices
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_maf")
    al = hal.Alignment.random(hal.RandOptions.preset("small", seed=0), device=0)
    g3 = al.genome_id("Genome_3")
    name, _, ln = al.sequences(g3)[0]
    got = al.maf_export(g3, 0, start=0, length=ln, unique=True, targets=[al.genome_id("Genome_1"), al.genome_id("Genome_2")])
    assert got == open(os.path.join(gold, "hal2mafMPTargetGenomesTest.maf")).read()
    g0 = al.genome_id("Genome_0")
    got = al.maf_export(g0, unique=True, ref_targets_bed=open(os.path.join(gold, "small-Genome_0.bed")).read())
    assert got == open(os.path.join(gold, "hal2mafMPRefTargetsGenomesTest.maf")).read()
    name, _, ln = al.sequences(g0)[0]
    got = _mp_slices(lambda s, l: al.maf_export(g0, 0, start=s, length=l, unique=True), name, ln, 250)
    assert got == open(os.path.join(gold, "hal2mafMPBySeqTest_Genome_0_seq.maf")).read()
    # and without --unique the slices of a run-compressed export still concatenate to the unsliced export's columns
    whole = al.maf_export(g0, 0)
    sliced = _mp_slices(lambda s, l: al.maf_export(g0, 0, start=s, length=l), name, ln, 250)
    cols = lambda t: sum(len(x.split("\t")[6]) for x in t.splitlines() if x.startswith("s\tGenome_0."))
    assert cols(whole) == cols(sliced) == ln


def test_reference_unit_tests_of_the_column_iterator(hal, oracle_bin, tmp_path):
    """The known answers of api/tests/halColumnIteratorTest.cpp (Depth, Dup, Inv) through hgx_column_rows, and row-for-row
    agreement with the oracle's ColumnIterator on the same alignments."""
    import handbuilt_columns as hc
    for name, build, check, refs in hc.CASES:
        img = str(tmp_path / (name + ".hgx"))
        halfix.write_hgx(img, build())
        al = hal.Alignment.open(img, device=0)
        for ref in refs:
            g = al.genome_id(ref)
            off, rows = al.column_rows(g, 0, 100)
            want = subprocess.run([oracle_bin, "columns", img, ref], check=True, stdout=subprocess.PIPE).stdout.decode().splitlines()
            for c in range(100):
                rs = rows[int(off[c]):int(off[c + 1])]
                tuples = [(al.genome_name(int(r["genome"])), int(r["pos"]), bool(r["reversed"])) for r in rs]
                check(ref, c, tuples)
                # the oracle prints ColumnMap order (by sequence), the rows API insertion order: same multiset
                o = sorted((x.split(":")[0], int(x.split(":")[1]), x.split(":")[2] == "-") for x in want[c].split()[1:])
                assert sorted(tuples) == o, (name, ref, c)


@pytest.mark.parametrize("seed", [2, 5, 6, 11])
def test_depth_by_tree_sweeps_vs_walk_and_oracle(hal, oracle_bin, tmp_path, seed, monkeypatch):
    """halAlignmentDepth by the two tree sweeps (HGX_DEPTH_SWEEP=1 lifts the size threshold) against the column walk (=0) and
    the oracle: every genome (leaves, ancestors, the root), whole genome and ranges, --step, --countDupes, --noAncestors,
    --targetGenomes."""
    al, img = _rand(hal, tmp_path, seed, dna=False, min_segments=120, max_segments=400)
    for g in range(al.num_genomes):
        n = al.genome_length(g)
        if n == 0:
            continue
        name = al.genome_name(g)
        leaf = not al.genome_children(g)
        others = [x for x in range(al.num_genomes) if x != g]
        cases = [dict(), dict(count_dupes=True), dict(targets=others[:2]), dict(targets=[others[-1]], count_dupes=True), dict(targets=others[1:4])]
        if leaf:
            cases += [dict(no_ancestors=True), dict(no_ancestors=True, targets=others[2:6])]
        for kw in cases:
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
            a = al.columns_depth(g, 0, n, **kw)
            b = al.columns_depth(g, n // 3, n // 2, **kw)
            c = al.columns_depth(g, 5, (n - 5) // 7, step=7, **kw)
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "0")
            assert np.array_equal(a, al.columns_depth(g, 0, n, **kw)), (name, kw)
            assert np.array_equal(b, al.columns_depth(g, n // 3, n // 2, **kw)), (name, kw)
            assert np.array_equal(c, al.columns_depth(g, 5, (n - 5) // 7, step=7, **kw)), (name, kw)
        monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        assert al.alignment_depth(g, count_dupes=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--countDupes"), name
        if leaf:
            assert al.alignment_depth(g, no_ancestors=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--noAncestors"), name
        assert al.alignment_depth(g, 0, start=7, length=n // 2, step=3) == \
            _oracle(oracle_bin, "depth", img, tmp_path, name, "--refSequence", al.sequences(g)[0][0], "--start", "7", "--length", str(n // 2), "--step", "3")


def test_column_tools_over_device_clones(hal, oracle_bin, tmp_path):
    """hgx_alignment_depth_multi / hgx_maf_export_multi over three handles of one alignment (clones on the test box's one GPU: the
    sharing out, the threads and the collation are the code that runs with three GPUs): the depth text is the single handle's, the
    sliced MAF is the expected file of the reference's own sliced runner (hal2mafMP.py, 250-base slices of Genome_0_seq with
    --unique, maf/Makefile:70-72), and the CLI twins take --devices."""
    from test_oracle_golden import _mp_slices
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_maf")
    al = hal.Alignment.random(hal.RandOptions.preset("small", seed=0), device=0)
    clones = [al, al.clone_to_device(0), al.clone_to_device(0)]
    g0 = al.genome_id("Genome_0")
    name, _, ln = al.sequences(g0)[0]
    want = open(os.path.join(gold, "hal2mafMPBySeqTest_Genome_0_seq.maf")).read()
    assert hal.maf_export_multi(clones, g0, 0, slice_size=250, unique=True) == want
    assert hal.maf_export_multi(clones[:1], g0, 0, slice_size=250, unique=True) == want
    # slice size 0: the range divided evenly over the handles, like hal2mafMP's default (ceil(length / numProc))
    size = -(-ln // 3)
    assert hal.maf_export_multi(clones, g0, 0) == _mp_slices(lambda s, l: al.maf_export(g0, 0, start=s, length=l), name, ln, size)
    for g in range(al.num_genomes):
        assert hal.alignment_depth_multi(clones, g) == al.alignment_depth(g)
        leaf = not al.genome_children(g)
        assert hal.alignment_depth_multi(clones, g, step=7, count_dupes=True, no_ancestors=leaf) == \
            al.alignment_depth(g, step=7, count_dupes=True, no_ancestors=leaf)
        sn, _, sl = al.sequences(g)[0]
        assert hal.maf_export_multi(clones, g, 0, slice_size=97, no_dupes=True) == \
            _mp_slices(lambda s, l: al.maf_export(g, 0, start=s, length=l, no_dupes=True), sn, sl, 97)
    # a larger alignment: the shares of a whole-genome scan go through the tree sweeps or the walk as their sizes say
    big, img = _rand(hal, tmp_path, 2, dna=False, min_segments=3000, max_segments=6000)
    bclones = [big, big.clone_to_device(0)]
    g9 = big.genome_id("Genome_9")
    assert hal.alignment_depth_multi(bclones, g9) == big.alignment_depth(g9) == _oracle(oracle_bin, "depth", img, tmp_path, "Genome_9")
    # CLI twins
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = str(tmp_path / "small.hgx")
    al.save(small)
    out = str(tmp_path / "d.wig")
    subprocess.check_call([os.path.join(root, "hal_amd", "_build", "halAlignmentDepth"), "--devices", "0,0", "--outWiggle", out, small, "Genome_0"])
    assert open(out).read() == al.alignment_depth(g0)
    out = str(tmp_path / "m.maf")
    subprocess.check_call([os.path.join(root, "hal_amd", "_build", "hal2maf"), "--devices", "0,0,0", "--sliceSize", "250", "--refGenome", "Genome_0",
                           "--refSequence", name, "--unique", small, out])
    assert open(out).read() == want


@pytest.mark.parametrize("seed,n_genomes,max_children,root_len", [(412, 8, 1, 1108), (77, 9, 2, 600), (5, 6, 3, 300)])
def test_maf_block_length_breaks_where_rows_change_sequence(hal, oracle_bin, tmp_path, seed, n_genomes, max_children, root_len):
    """A block-length limit that falls into a run of columns makes the host go through the per-column logic in the middle of the run,
    with the run's rows moved on — where a row has just crossed into its genome's next sequence its sequence's bases must still
    come in the walk's order (seed 412: a soak run found two rows of one sequence swapped there).  Every genome as reference."""
    import halfix
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=n_genomes, max_children=max_children, root_len=root_len))
    al = hal.Alignment.open(img, device=0)
    for g in range(al.num_genomes):
        if al.genome_length(g) == 0:
            continue
        nm = al.genome_name(g)
        for mbl in (17, 3):
            assert al.maf_export(g, max_block_len=mbl) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--maxBlockLen", str(mbl)), (nm, mbl)


def test_maf_print_tree_against_the_oracle(hal, oracle_bin, tmp_path):
    """hal2maf --printTree (maf/impl/halMafBlock.cpp:121-292, 443-448, 485-497): every block with the tree of its rows, blocks also
    ending where the column's tree changes, rows in the tree's post order.  The reference holds no expected file for the option
    and its tree code (sonLib) is not in the reference tree: held against the oracle's restatement only.  Where the reference has
    no tree for a column (its first base an insertion in a genome with bottom segments; a paralog --noDupes left out) the library
    and the oracle refuse with the same words."""
    import halfix
    import make_maf_batches as mk
    trees = refused = 0
    images = []
    for seed in (2, 4):
        al, img = _rand(hal, tmp_path, seed)
        images.append((al, img))
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(3, n_genomes=5, max_children=2, root_len=150))
    images.append((hal.Alignment.open(img, device=0), img))
    for al, img in images:
        for g in range(al.num_genomes):
            if al.genome_length(g) == 0:
                continue
            nm = al.genome_name(g)
            for kw, args in ((dict(), []), (dict(max_block_len=7), ["--maxBlockLen", "7"]), (dict(only_orthologs=True), ["--onlyOrthologs"])):
                out = str(tmp_path / "o.maf")
                r = subprocess.run([oracle_bin, "maf", img, out, "--refGenome", nm, "--printTree"] + args, stderr=subprocess.PIPE)
                try:
                    got = al.maf_export(g, print_tree=True, **kw)
                except hal.HgxError as e:
                    assert r.returncode != 0 and any(m in str(e) and m in r.stderr.decode() for m in mk.NO_TREE), (nm, kw, str(e))
                    refused += 1
                    continue
                assert r.returncode == 0 and got == open(out).read(), (nm, kw)
                assert 'a tree="' in got
                trees += 1
    assert trees >= 6 and refused >= 1
