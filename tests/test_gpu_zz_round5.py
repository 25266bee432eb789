"""Round 5's GPU tests, in a file of their own that sorts behind the others: hal2maf's device stage from per-base tracks
(hal_amd/csrc/hgx_maf_kernels.hpp) forced on and held against the column walk and the oracle; the tree sweeps with sums over a
polytomy; halAlignmentDepth's wig through several chunks; the full-size 50-genome alignment against the oracle itself."""
import os
import subprocess
import sys

import numpy as np
import pytest

import halfix
from util import oracle_liftover

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_queries, workload_options  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- helpers ----


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


def _both_ways(al, monkeypatch, *args, **kw):
    """the export with the heads from the per-base tracks (forced: a difference from the walk in the first chunk, or sizes that do
    not add up, raise) and by the column walk: the same text; the tracks were used and checked"""
    monkeypatch.setenv("HGX_MAF_SWEEP", "1")
    al.maf_tracks_info(drop=True)
    a = al.maf_export(*args, **kw)
    info = al.maf_tracks_info()
    assert info["tracks"] and info["state"].startswith("checked") and info["chunks_served"] >= 1, info
    monkeypatch.setenv("HGX_MAF_SWEEP", "0")
    b = al.maf_export(*args, **kw)
    monkeypatch.delenv("HGX_MAF_SWEEP")
    assert a == b
    return a


def _unique_both_ways(al, monkeypatch, *args, **kw):
    monkeypatch.setenv("HGX_MAF_SWEEP", "1")
    al.maf_tracks_info(drop=True)
    a = al.maf_export(*args, unique=True, **kw)
    info = al.maf_tracks_info()
    assert info["tracks"] and info["state_unique"].startswith("checked") and info["chunks_served_unique"] >= 1, info
    monkeypatch.setenv("HGX_MAF_SWEEP", "0")
    b = al.maf_export(*args, unique=True, **kw)
    monkeypatch.delenv("HGX_MAF_SWEEP")
    assert a == b
    return a


def _bed(seq_name, starts, lens, strand, lo, hi):
    return "".join("%s\t%d\t%d\tq\t0\t%s\n" % (seq_name, int(starts[i]), int(starts[i] + lens[i]), chr(int(strand[i]))) for i in range(lo, hi))


# ---- what the round-4 review asked for and no new device code is needed for: the sums' tail (ADVICE), the wig's chunks, config 4 at full size ----


def test_count_dupes_sweep_over_a_polytomy_with_segment_tails(hal, oracle_bin, tmp_path, monkeypatch):
    """--countDupes by the tree sweeps when a genome has more than eight children in scope (k_sweep_up runs once per eight and
    adds to what the launch before left) and bottom segments whose length leaves one base past a round of lanes (33, 65, 97):
    the lane at a segment's end must not add a launch's children twice to a base its neighbour of the round before has stored."""
    import random
    rnd = random.Random(3)
    lens = [33, 65, 33, 97, 1, 34, 65, 129, 2, 33]
    starts = [sum(lens[:i]) for i in range(len(lens))]
    total = sum(lens)
    nkids = 12
    genomes = [None] * (nkids + 1)
    slots = []
    for c in range(nkids):
        tops, pos = [], 0
        members = {}
        order = list(range(len(lens))) + [rnd.randrange(len(lens)) for _ in range(4)]  # (every segment once, four of them twice)
        rnd.shuffle(order)
        for k, j in enumerate(order):
            tops.append([pos, lens[j], j, rnd.random() < 0.4, -1])
            members.setdefault(j, []).append(k)
            pos += lens[j]
        for j, ms in members.items():
            if len(ms) > 1:
                for a, b in zip(ms, ms[1:] + ms[:1]):
                    tops[a][4] = b
        genomes[c + 1] = halfix.simple_genome("L%d" % c, 0, [], pos, [tuple(t) for t in tops], [], seqname="L%d_chr" % c)
        slots.append({j: (ms[-1], tops[ms[-1]][3]) for j, ms in members.items()})
    bots = [(starts[j], lens[j], [slots[c][j] for c in range(nkids)]) for j in range(len(lens))]
    genomes[0] = halfix.simple_genome("Root", -1, list(range(1, nkids + 1)), total, [], bots, seqname="Root_chr")
    img = str(tmp_path / "poly.hgx")
    halfix.write_hgx(img, genomes)
    al = hal.Alignment.open(img, device=0)
    for g in (0, 1, nkids):
        name, n = al.genome_name(g), al.genome_length(g)
        for kw in (dict(count_dupes=True), dict()):
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
            a = al.columns_depth(g, 0, n, **kw)
            monkeypatch.setenv("HGX_DEPTH_SWEEP", "0")
            assert np.array_equal(a, al.columns_depth(g, 0, n, **kw)), (name, kw)
        monkeypatch.setenv("HGX_DEPTH_SWEEP", "1")
        assert al.alignment_depth(g, count_dupes=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--countDupes"), name
        assert al.alignment_depth(g) == _oracle(oracle_bin, "depth", img, tmp_path, name), name
        assert _both_ways(al, monkeypatch, g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), name


@pytest.mark.parametrize("chunk", [97, 1000, 4099])
def test_depth_wig_through_several_chunks(hal, oracle_bin, tmp_path, monkeypatch, chunk):
    """hgx_alignment_depth hands its values to the line writers chunk by chunk through two page-locked blocks (sixteen million
    columns a chunk: HGX_WIG_CHUNK sets it): genomes of a few thousand columns through 3 .. a hundred chunks — the device copies,
    the hand-off and the lines in place — against the oracle's wig and the device values."""
    al, img = _rand(hal, tmp_path, 6, dna=False, min_segments=240, max_segments=400)
    monkeypatch.setenv("HGX_WIG_CHUNK", str(chunk))
    for g in (al.num_genomes - 1, 0, 3):
        n, name = al.genome_length(g), al.genome_name(g)
        assert n >= 2 * chunk
        want = _oracle(oracle_bin, "depth", img, tmp_path, name)
        assert al.alignment_depth(g) == want, (name, chunk)
        vals = al.columns_depth(g, 0, n)
        assert [int(x) for x in want.split("\n")[1:-1]] == vals.tolist()
        assert al.alignment_depth(g, count_dupes=True) == _oracle(oracle_bin, "depth", img, tmp_path, name, "--countDupes"), (name, chunk)
        assert al.alignment_depth(g, 0, start=7, length=n // 2, step=3) == \
            _oracle(oracle_bin, "depth", img, tmp_path, name, "--refSequence", al.sequences(g)[0][0], "--start", "7", "--length", str(n // 2),
                    "--step", "3")


def test_config4_full_size_sample_vs_oracle(hal, oracle_bin, tmp_path):
    """The FULL-size 50-genome alignment against the oracle itself: the first 5 000 intervals of the shard's batch Genome_44 ->
    Genome_2 (the default plan: merged table, general intervals, the LDS finishing kernels for the sets of more than 64 pieces),
    then 1 500 of them without dupes, and halAlignmentDepth of a 100 k-column window."""
    al = hal.Alignment.random(workload_options(1.0, "cfg4"), device=0)
    src, tgt = al.genome_id("Genome_44"), al.genome_id("Genome_2")
    seq_name, seq_start, length = al.sequences(src)[0]
    assert al.num_genomes == 50 and length > 50_000_000
    starts, lens, strand = make_queries(length, 1250000, 1234)
    img = str(tmp_path / "cfg4.hgx")
    al.save(img)
    bed = _bed(seq_name, starts, lens, strand, 0, 5000)
    got = hal.liftover_convert(al, src, bed, tgt)
    assert got == oracle_liftover(oracle_bin, img, "Genome_44", "Genome_2", bed, tmp_path)
    assert got.count("\n") > 10 * 5000
    bed2 = _bed(seq_name, starts, lens, strand, 5000, 6500)
    assert hal.liftover_convert(al, src, bed2, tgt, traverse_dupes=False) == \
        oracle_liftover(oracle_bin, img, "Genome_44", "Genome_2", bed2, tmp_path, no_dupes=True)
    a, ln = length // 2, 100000
    wig = str(tmp_path / "o.wig")
    subprocess.check_call([oracle_bin, "depth", img, "Genome_44", wig, "--refSequence", seq_name, "--start", str(a), "--length", str(ln)])
    assert al.alignment_depth(src, 0, start=a, length=ln) == open(wig).read()


# ---- several writers, the tools with one process per GPU ----


def test_writers_group_through_the_c_abi(hal, tmp_path):
    """several writers (hgx_liftover_gather_writers; with the test box's one rank the group is the rank itself) and the sizes that
    place their texts (hgx_comm_all_sizes): the writer's blob is the blob the root of hgx_liftover_gather gets"""
    import torch
    from hal_amd import shard
    from test_gpu_exchange import _batch
    from test_gpu_liftover import _rand_alignment
    al, _ = _rand_alignment(hal, tmp_path, 2)
    src, tgt, gs, ge, st = _batch(hal, al, 3000, 5)
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=3000)
    comm = hal.Comm(hal.Comm.unique_id(), 0, 1, 0)
    plan.run(gs, ge, st)
    slot = (plan.wire_capacity() + 7) // 8 * 8
    gx = shard.SlotExchange(1, 0, slot, "cuda", backend="c_abi", comm=comm, root=0, bed_only=True)
    gx.submit(plan, first_query=12345)
    (blob8,) = gx.slots(gx.wait())
    torch.cuda.synchronize()
    for group in (1, 2):
        wx = shard.SlotExchange(1, 0, slot, "cuda", backend="c_abi", comm=comm, group=group, bed_only=True)
        wx.submit(plan, first_query=12345)
        buf = wx.wait()
        torch.cuda.synchronize()
        (blobw,) = wx.slots(buf)
        assert wx.last_bytes == blobw.numel() and torch.equal(blobw.cpu(), blob8.cpu())
    assert comm.all_sizes(123456789012) == [123456789012]
    comm.close()


def test_liftover_over_the_ranks_of_a_node_every_rank_a_writer(hal, oracle_bin, tmp_path):
    """hal_amd.liftover_mp (one process per GPU, the ranks write their shares side by side) with the test box's one rank: the
    file is halLiftover's; and through the launcher with two processes on this one GPU"""
    import socket
    import torch.distributed as dist
    from hal_amd import liftover_mp
    from test_gpu_liftover import _rand_alignment
    al, img = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    seq, _, n = al.sequences(src)[0]
    rng = np.random.default_rng(9)
    lines = []
    for i in range(4000):
        a = int(rng.integers(0, n - 300))
        lines.append("%s\t%d\t%d\tn%d\t0\t%s" % (seq, a, a + int(rng.integers(1, 300)), i, "+-"[i & 1]))
    bed = str(tmp_path / "in.bed")
    open(bed, "w").write("\n".join(lines) + "\n")
    want = hal.liftover_convert(al, src, open(bed).read(), tgt)
    assert want == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", open(bed).read(), tmp_path)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        out = str(tmp_path / "out1.bed")
        assert liftover_mp.run(img, "Genome_9", bed, "Genome_2", out, device=0) == len(want.encode())
        assert open(out).read() == want
    finally:
        dist.destroy_process_group()
    out2 = str(tmp_path / "out2.bed")
    env = dict(os.environ, HGX_MP_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "hal_amd.liftover_mp", img, "Genome_9", bed, "Genome_2", out2],
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert open(out2).read() == want


# ---- hal2maf's block state machine over slices of the export (host code) ----


def test_maf_walk_over_slices_at_full_size(hal, oracle_bin, tmp_path, monkeypatch):
    """the block state machine over slices of the export side by side (HGX_MAF_SLICED=1: whatever the host's size) on config 3's
    alignment: 4 M columns in batches of 250 k (16 slices) give the one-thread walk's text, with and without --unique, and a
    300 k-column stretch of it is the oracle's"""
    al = hal.Alignment.random(workload_options(1.0, "cfg2", dna="fast"), device=0)
    src = al.genome_id("Genome_9")
    seq = al.sequences(src)[0][0]
    a, ln = 23_000_000, 4_000_000
    monkeypatch.setenv("HGX_MAF_CHUNK", "250000")
    for kw in (dict(), dict(unique=True)):
        monkeypatch.setenv("HGX_MAF_SLICED", "1")
        sliced = al.maf_export(src, 0, start=a, length=ln, no_ancestors=True, **kw)
        info = al.maf_tracks_info()["last_export"]
        assert info["walk"].startswith("slices") and info["slices"] == 16 and info["rounds"] >= 1, info
        monkeypatch.setenv("HGX_MAF_SLICED", "0")
        assert sliced == al.maf_export(src, 0, start=a, length=ln, no_ancestors=True, **kw), kw
        assert al.maf_tracks_info()["last_export"]["walk"] == "one thread"
    img = str(tmp_path / "cfg2.hgx")
    al.save(img)
    monkeypatch.setenv("HGX_MAF_SLICED", "1")
    monkeypatch.setenv("HGX_MAF_CHUNK", "20000")
    got = al.maf_export(src, 0, start=a, length=300000, no_ancestors=True)
    assert got == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", "Genome_9", "--noAncestors", "--refSequence", seq, "--start", str(a),
                          "--length", "300000")


# ---- hal2maf's heads from per-base tracks (round 5's device code: last, so that nothing above waits behind it) ----


def test_maf_tracks_reference_goldens(hal, monkeypatch):
    """the reference's own expected files (maf/tests/expected) through the tracks"""
    al = hal.Alignment.random(hal.RandOptions.preset("small", seed=0), device=0)
    assert _both_ways(al, monkeypatch, al.genome_id("Genome_0")) == open(os.path.join(GOLD, "ref_maf", "hal2mafSmallTest.maf")).read()
    g2 = al.genome_id("Genome_2")
    assert _both_ways(al, monkeypatch, g2, 0, start=1000, length=2000) == open(os.path.join(GOLD, "ref_maf", "hal2mafSeqPartTest.maf")).read()


@pytest.mark.parametrize("seed", [2, 5, 6, 11])
def test_maf_tracks_vs_walk_and_oracle(hal, oracle_bin, tmp_path, seed, monkeypatch):
    """random alignments with inversions and paralogy rings: every genome as reference, --noAncestors on leaves, target sets,
    ranges; device batches of 13 and 1000 columns (the tracks serve every chunk; marked columns at chunk ends)"""
    al, img = _rand(hal, tmp_path, seed)
    for chunk in ("13", "1000"):
        monkeypatch.setenv("HGX_MAF_CHUNK", chunk)
        for g in range(al.num_genomes):
            n, name = al.genome_length(g), al.genome_name(g)
            if n == 0 or (chunk == "13" and g % 3 != seed % 3):
                continue
            leaf = not al.genome_children(g)
            others = [x for x in range(al.num_genomes) if x != g]
            assert _both_ways(al, monkeypatch, g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), (name, chunk)
            if leaf:
                assert _both_ways(al, monkeypatch, g, no_ancestors=True) == \
                    _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--noAncestors"), (name, chunk)
            tg = others[1:4]
            assert _both_ways(al, monkeypatch, g, targets=tg) == \
                _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--targetGenomes", ",".join(al.genome_name(x) for x in tg)), (name, chunk)
            seq = al.sequences(g)[0][0]
            assert _both_ways(al, monkeypatch, g, 0, start=n // 3, length=n // 2, max_block_len=50) == \
                _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--refSequence", seq, "--start", str(n // 3), "--length", str(n // 2),
                        "--maxBlockLen", "50"), (name, chunk)


@pytest.mark.parametrize("seed", [2, 11])
def test_maf_tracks_no_dupes(hal, oracle_bin, tmp_path, seed, monkeypatch):
    """--noDupes through the tracks (round 6): no paralogy ring is followed and only the segment a parent's slot names goes up
    (halColumnIterator.cpp:560, 645) — the sweeps, the break sweeps and the row selection take the option as a flag; with --unique,
    target sets and ranges, against the column walk and the oracle"""
    al, img = _rand(hal, tmp_path, seed)
    for chunk in ("97", "5000"):
        monkeypatch.setenv("HGX_MAF_CHUNK", chunk)
        for g in range(al.num_genomes):
            n, name = al.genome_length(g), al.genome_name(g)
            if n == 0 or (chunk == "97" and g % 3 != seed % 3):
                continue
            others = [x for x in range(al.num_genomes) if x != g]
            assert _both_ways(al, monkeypatch, g, no_dupes=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--noDupes"), (name, chunk)
            assert _unique_both_ways(al, monkeypatch, g, no_dupes=True) == \
                _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--noDupes", "--unique"), (name, chunk)
            tg = others[1:4]
            seq = al.sequences(g)[0][0]
            assert _both_ways(al, monkeypatch, g, 0, start=n // 3, length=n // 2, max_block_len=50, targets=tg, no_dupes=True) == \
                _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--refSequence", seq, "--start", str(n // 3), "--length", str(n // 2),
                        "--maxBlockLen", "50", "--noDupes", "--targetGenomes", ",".join(al.genome_name(x) for x in tg)), (name, chunk)
    star = halfix.random_multiseq_alignment(5, n_genomes=41, max_children=3, root_len=300, root_children=24)
    simg = str(tmp_path / "star.hgx")
    halfix.write_hgx(simg, star)
    sal = hal.Alignment.open(simg, device=0)
    monkeypatch.delenv("HGX_MAF_CHUNK")
    for g in (0, 7, sal.num_genomes - 1):
        if sal.genome_length(g):
            assert _both_ways(sal, monkeypatch, g, no_dupes=True) == \
                _oracle(oracle_bin, "maf", simg, tmp_path, "--refGenome", sal.genome_name(g), "--noDupes"), g


def test_maf_tracks_multiseq_and_star(hal, oracle_bin, tmp_path, monkeypatch):
    """the independent generator's alignments (several sequences a genome, irregular segments, insertions, deletions) and a star of
    twenty-four children under the root (three launches of the break sweep for the root)"""
    cases = [("m1", halfix.random_multiseq_alignment(1, n_genomes=8, max_children=3, root_len=400)),
             ("m6", halfix.random_multiseq_alignment(6, n_genomes=9, max_children=3, root_len=300)),
             ("star", halfix.random_multiseq_alignment(5, n_genomes=41, max_children=3, root_len=300, root_children=24))]
    for tag, genomes in cases:
        img = str(tmp_path / (tag + ".hgx"))
        halfix.write_hgx(img, genomes)
        al = hal.Alignment.open(img, device=0)
        refs = range(al.num_genomes) if tag != "star" else (0, 1, 7, 25, al.num_genomes - 1)
        for g in refs:
            if al.genome_length(g) == 0:
                continue
            name = al.genome_name(g)
            assert _both_ways(al, monkeypatch, g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), (tag, name)
            if not al.genome_children(g):
                assert _both_ways(al, monkeypatch, g, no_ancestors=True, only_sequence_names=True) == \
                    _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name, "--noAncestors", "--onlySequenceNames"), (tag, name)


def test_maf_tracks_on_int64_tables_and_real_data(hal, oracle_bin, tmp_path, monkeypatch):
    al, img = _rand(hal, tmp_path, 2)
    monkeypatch.setenv("HGX_FORCE_WIDE", "1")
    wide = al.clone_to_device(0)
    monkeypatch.delenv("HGX_FORCE_WIDE")
    for g in (al.num_genomes - 1, 0, 3):
        name = al.genome_name(g)
        assert _both_ways(wide, monkeypatch, g) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name), name
    mr = os.path.join(GOLD, "ref_hdf5", "mr.hal")
    real = hal.Alignment.open(mr, device=0)
    rimg = str(tmp_path / "mr.hgx")
    real.save(rimg)
    for g in range(real.num_genomes):
        name = real.genome_name(g)
        seq, _, n = real.sequences(g)[0]
        ln = min(n - n // 5, 300000)  # start + length stays inside the sequence (maf/impl/halMafExport.cpp:25-37)
        assert _both_ways(real, monkeypatch, g, 0, start=n // 5, length=ln) == \
            _oracle(oracle_bin, "maf", rimg, tmp_path, "--refGenome", name, "--refSequence", seq, "--start", str(n // 5), "--length", str(ln)), name


def test_maf_tracks_unique_small(hal, oracle_bin, tmp_path, monkeypatch):
    """a small case of test_maf_tracks_unique (the CPU suite runs it on the host-side emulation, tests/test_cpu_emulation.py)"""
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(2, n_genomes=4, root_len=200))
    al = hal.Alignment.open(img, device=0)
    monkeypatch.setenv("HGX_MAF_CHUNK", "150")
    n_bytes = 0
    for g in range(al.num_genomes):
        nm = al.genome_name(g)
        if al.genome_length(g) == 0:
            continue
        want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique")
        assert _unique_both_ways(al, monkeypatch, g) == want, nm
        n_bytes += len(want)
        sname, _, slen = al.sequences(g)[-1]
        if slen >= 4 and g == al.num_genomes - 1:
            assert _unique_both_ways(al, monkeypatch, g, len(al.sequences(g)) - 1, start=1, length=slen - 1) == \
                _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--refSequence", sname, "--start", "1", "--length", str(slen - 1),
                        "--unique"), (nm, sname)
    assert n_bytes > 0
    # batches with a column of more reference copies than a lane holds go to the walk, the others are served (the limit lowered to
    # none: every batch with a paralog of the reference)
    monkeypatch.setenv("HGX_MAF_UNIQUE_MAX_REF", "0")
    monkeypatch.setenv("HGX_MAF_SWEEP", "1")
    for g in (1, al.num_genomes - 1):
        if al.genome_length(g):
            al.maf_tracks_info(drop=True)
            assert al.maf_export(g, unique=True) == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", al.genome_name(g), "--unique")


@pytest.mark.parametrize("seed", [0, 3, 4])
def test_maf_tracks_unique(hal, oracle_bin, tmp_path, seed, monkeypatch):
    """--unique from the marked columns' rows (the stretches of a run that are passed over, walked for their keys, written): several
    sequences a genome (a sequence's range begins inside the genome: paralogs left of it), every genome as reference, whole
    sequences and ranges, device batches of 13 columns and whole, --noAncestors, target sets"""
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=6))
    al = hal.Alignment.open(img, device=0)
    n_bytes = 0
    for chunk in (None, "13"):
        if chunk:
            monkeypatch.setenv("HGX_MAF_CHUNK", chunk)
        for g in range(al.num_genomes):
            nm = al.genome_name(g)
            if al.genome_length(g) == 0:
                continue
            want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique")
            assert _unique_both_ways(al, monkeypatch, g) == want, (nm, chunk)
            n_bytes += len(want)
            if not al.genome_children(g):
                assert _unique_both_ways(al, monkeypatch, g, no_ancestors=True) == \
                    _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique", "--noAncestors"), (nm, chunk)
            tg = [x for x in range(al.num_genomes) if x != g][1:4]
            assert _unique_both_ways(al, monkeypatch, g, targets=tg) == \
                _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--unique", "--targetGenomes",
                        ",".join(al.genome_name(x) for x in tg)), (nm, chunk)
            for si, (sname, _, slen) in enumerate(al.sequences(g)):
                if slen < 4:
                    continue
                for a, ln in ((slen // 3, slen // 2), (1, slen - 1)):
                    assert _unique_both_ways(al, monkeypatch, g, si, start=a, length=ln) == \
                        _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", nm, "--refSequence", sname, "--start", str(a), "--length", str(ln),
                                "--unique"), (nm, sname, a, ln, chunk)
    assert n_bytes > 0
    # the generator of the reference's own tests: inversions and paralogy rings inside one sequence
    al2, img2 = _rand(hal, tmp_path, seed + 1)
    monkeypatch.setenv("HGX_MAF_CHUNK", "50")
    for g in range(al2.num_genomes):
        nm = al2.genome_name(g)
        assert _unique_both_ways(al2, monkeypatch, g) == _oracle(oracle_bin, "maf", img2, tmp_path, "--refGenome", nm, "--unique"), nm


def test_maf_tracks_at_full_size(hal, oracle_bin, tmp_path, monkeypatch):
    """config 3's alignment at full size: 3 M columns by the tracks and by the walk (the same text), a 200 k-column slice of them
    against the oracle, and the tracks kept with the handle serve the second export without being built again"""
    al = hal.Alignment.random(workload_options(1.0, "cfg2", dna="fast"), device=0)
    src = al.genome_id("Genome_9")
    seq = al.sequences(src)[0][0]
    a, ln = 11_000_000, 3_000_000
    text = _both_ways(al, monkeypatch, src, 0, start=a, length=ln, no_ancestors=True)
    assert text.count("\na") > ln // 200
    built = al.maf_tracks_info()
    img = str(tmp_path / "cfg2.hgx")
    al.save(img)
    monkeypatch.setenv("HGX_MAF_SWEEP", "1")
    got = al.maf_export(src, 0, start=a + 1_000_000, length=200000, no_ancestors=True)
    after = al.maf_tracks_info()
    assert after["build_ms"] == built["build_ms"] and after["chunks_served"] > built["chunks_served"]
    assert got == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", "Genome_9", "--noAncestors", "--refSequence", seq, "--start",
                          str(a + 1_000_000), "--length", "200000")
    # --unique from the same tracks: the range begins inside the genome (copies of its bases left of it: columns walked for their
    # keys only), the generator's paralogs inside it (columns passed over)
    text_u = _unique_both_ways(al, monkeypatch, src, 0, start=a, length=ln, no_ancestors=True)
    assert text_u.count("\na") > ln // 400  # (columns passed over cut blocks in two: no bound by the plain export's size)
    monkeypatch.setenv("HGX_MAF_SWEEP", "1")
    got_u = al.maf_export(src, 0, start=a + 1_000_000, length=200000, no_ancestors=True, unique=True)
    assert got_u == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", "Genome_9", "--noAncestors", "--unique", "--refSequence", seq,
                            "--start", str(a + 1_000_000), "--length", "200000")
    # --noDupes through the tracks (round 6; another set of tracks: the option is part of what they were built for)
    text_n = _both_ways(al, monkeypatch, src, 0, start=a, length=1_000_000, no_ancestors=True, no_dupes=True)
    assert text_n.count("\na") > 1000
    monkeypatch.setenv("HGX_MAF_SWEEP", "1")
    got_n = al.maf_export(src, 0, start=a + 500_000, length=100000, no_ancestors=True, no_dupes=True)
    assert got_n == _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", "Genome_9", "--noAncestors", "--noDupes", "--refSequence", seq,
                            "--start", str(a + 500_000), "--length", "100000")
    # ... and over the slices of hgx_maf_export_multi (an export each; the handle's tracks serve all of them)
    multi = hal.maf_export_multi([al], src, 0, start=a, length=ln, slice_size=500000, no_ancestors=True, unique=True)
    monkeypatch.setenv("HGX_MAF_SWEEP", "0")
    assert multi == hal.maf_export_multi([al], src, 0, start=a, length=ln, slice_size=500000, no_ancestors=True, unique=True)


def test_export_multi_again_and_again(hal):
    """hgx_maf_export_multi's slices run several at a time (four a device): texts that grow by mremap side by side (textRealloc's
    registry: round 6 found a moving block's old address erased after another thread had been handed it) and the null-stream path of
    a one-batch export's heads, which is entered one chunk at a time.  Twelve passes over a genome of 5 M columns in slices of 100 k,
    two handles of this GPU: the same text every time, and the text of one slice at a time a handle."""
    al = hal.Alignment.random(workload_options(0.1, "cfg2", dna="fast"), device=0)
    src = al.genome_id("Genome_9")
    n = al.genome_length(src)
    clones = [al, al.clone_to_device(0)]
    os.environ["HGX_MAF_MULTI_PER_HANDLE"] = "1"
    try:
        want = hal.maf_export_multi([al], src, 0, start=0, length=n, slice_size=100000, no_ancestors=True, unique=True)
    finally:
        del os.environ["HGX_MAF_MULTI_PER_HANDLE"]
    assert want.count("\na") > n // 400
    for rep in range(12):
        if rep % 3 == 0:  # (the tracks built again, by the first slices of a pass, side by side on the two handles)
            for c in clones:
                c.maf_tracks_info(drop=True)
        assert hal.maf_export_multi(clones, src, 0, start=0, length=n, slice_size=100000, no_ancestors=True, unique=True) == want, rep


def test_hal2maf_over_the_ranks_of_a_node_every_rank_a_writer(hal, tmp_path):
    """hal_amd.maf_mp (hal2mafMP.py's slices, a contiguous run of them a rank, the ranks write side by side): the file is
    hgx_maf_export_multi's with the same slice size — one rank in process, two and three processes through the launcher on this one
    GPU; several sequences, --unique, a sub-range"""
    import socket
    import torch.distributed as dist
    from hal_amd import maf_mp
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(4, n_genomes=6))
    al = hal.Alignment.open(img, device=0)
    ref = al.num_genomes - 1
    name = al.genome_name(ref)
    want_all = hal.maf_export_multi([al], ref, -1, slice_size=37, unique=True)
    seq, _, slen = max(al.sequences(ref), key=lambda t: t[2])
    si = [t[0] for t in al.sequences(ref)].index(seq)
    want_part = hal.maf_export_multi([al], ref, si, start=3, length=slen - 5, slice_size=11, no_dupes=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        out = str(tmp_path / "one.maf")
        assert maf_mp.run(img, out, ref_genome=name, slice_size=37, device=0, unique=True) == len(want_all.encode())
        assert open(out).read() == want_all
    finally:
        dist.destroy_process_group()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for nproc, args, want in ((2, ["--refGenome", name, "--sliceSize", "37", "--unique"], want_all),
                              (3, ["--refGenome", name, "--refSequence", seq, "--refStart", "3", "--length", str(slen - 5), "--sliceSize", "11", "--noDupes"],
                               want_part)):
        out = str(tmp_path / ("mp%d.maf" % nproc))
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                            "--master-port", str(port), "-m", "hal_amd.maf_mp", img, out] + args,  # (the launcher takes what follows the module for its own until a positional)
                           cwd=root, env=dict(os.environ, HGX_MP_DEVICE="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert open(out).read() == want, (nproc, args)


# ---- round 6: the batches as a stream, the text rendered on the device, a writer's blobs rendered in the library ----


def test_maf_stream_and_device_render_against_the_host_paths(hal, oracle_bin, tmp_path, monkeypatch):
    """the plain export three ways — batches as a stream + text rendered on the device (the defaults), the launches that copy
    (HGX_MAF_STREAM=0) + the host's rendering threads (HGX_MAF_DEVICE_RENDER=0), and a stream whose buffers have no room for its
    batches (it hands them to the other launches) — give the oracle's text; chunks of 777 columns so that every path crosses batch
    ends and the stream's eighth batches are held against the walk"""
    al, img = _rand(hal, tmp_path, 4, min_segments=300, max_segments=500)
    monkeypatch.setenv("HGX_MAF_SWEEP", "1")
    monkeypatch.setenv("HGX_MAF_DEVICE_RENDER_MIN_BLOCKS", "1")
    for g in (al.num_genomes - 1, 0):
        name = al.genome_name(g)
        want = _oracle(oracle_bin, "maf", img, tmp_path, "--refGenome", name)
        for chunk in ("777", "100000"):
            monkeypatch.setenv("HGX_MAF_CHUNK", chunk)
            al.maf_tracks_info(drop=True)
            assert al.maf_export(g) == want, (name, chunk, "stream + device render")
            served = al.maf_tracks_info()["chunks_served"]
            assert served >= 1
            monkeypatch.setenv("HGX_MAF_STREAM", "0")
            monkeypatch.setenv("HGX_MAF_DEVICE_RENDER", "0")
            assert al.maf_export(g) == want, (name, chunk, "copying launches + rendering threads")
            monkeypatch.delenv("HGX_MAF_STREAM")
            monkeypatch.delenv("HGX_MAF_DEVICE_RENDER")
            monkeypatch.setenv("HGX_MAF_STREAM_ROOM", "3")  # (three rows of room: no batch fits)
            assert al.maf_export(g) == want, (name, chunk, "a stream without room")
            monkeypatch.delenv("HGX_MAF_STREAM_ROOM")
    monkeypatch.delenv("HGX_MAF_CHUNK")


def test_a_writers_blobs_rendered_in_the_library(hal, oracle_bin, tmp_path):
    """hgx_liftover_render_blobs on the device's own wire blobs (12-byte and 8-byte forms, two shards as two ranks would lift them):
    the text is hgx_liftover_convert's, which is the oracle's"""
    from test_gpu_liftover import _rand_alignment
    import torch
    al, img = _rand_alignment(hal, tmp_path, 2)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_2")
    seq, ss, n = al.sequences(src)[0]
    rng = np.random.default_rng(11)
    starts = rng.integers(0, n - 300, 3000)
    lens = rng.integers(1, 300, 3000)
    lines = ["%s\t%d\t%d\tn%d\t%d\t%s\n" % (seq, starts[i], starts[i] + lens[i], i, i % 7, "+-."[i % 3]) for i in range(3000)]
    want = hal.liftover_convert(al, src, "".join(lines), tgt)
    assert want == oracle_liftover(oracle_bin, img, "Genome_9", "Genome_2", "".join(lines), tmp_path)
    gs = torch.tensor(starts + ss, dtype=torch.int64, device="cuda")
    ge = torch.tensor(starts + lens - 1 + ss, dtype=torch.int64, device="cuda")
    st = torch.tensor([ord("+-."[i % 3]) for i in range(3000)], dtype=torch.uint8, device="cuda")
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=3000)
    for bed_only in (False, True):
        blobs = []
        for lo, hi in ((0, 1100), (1100, 3000)):
            plan.run(gs[lo:hi].contiguous(), ge[lo:hi].contiguous(), st[lo:hi].contiguous())
            blob, fmt = plan.wire_blob(first_query=lo, bed_only=bed_only)
            assert fmt == (8 if bed_only else 12)
            blobs.append(blob.cpu())
        assert hal.liftover_render_blobs(al, src, tgt, "".join(lines), blobs).decode() == want
