"""halGetBlocksInTargetRange behind the C ABI (hgx_get_blocks_in_target_range[s]: BlockMapper with adjacencies mapped back,
paralogy chaining, fragment merging, target dupe lists; blockViz/impl/halBlockViz.cpp:243-330, 759-1178,
liftover/impl/halBlockMapper.cpp:36-245) against the reference's own expected output and against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import halfix
from test_gpu_liftover import _rand_alignment

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _oracle(oracle_bin, img, q, t, chrom, start, end, seq=False, dup_mode=2, adj=True, limit=None, reversed_=False):
    cmd = [oracle_bin, "blockviz", img, q, t, chrom, str(start), str(end), "--dupMode", str(dup_mode)]
    if seq:
        cmd.append("--doSeq")
    if not adj:
        cmd.append("--noAdj")
    if reversed_:
        cmd.append("--tReversed")
    if limit:
        cmd += ["--coalescenceLimit", limit]
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE).stdout.decode()


def test_reference_expected_output(hal, tmp_path):
    """blockViz/Makefile:52-71: blockVizTest --verbose --doSeq ... Genome_2 Genome_0 Genome_0_seq 0 3000 on halRandGen --preset small
    --seed 0 --minSegmentLength 3000 --maxSegmentLength 5000, through the HIP path"""
    al = hal.Alignment.random(hal.RandOptions.preset("small", seed=0, min_segment_length=3000, max_segment_length=5000), device=0)
    blocks, dupes = al.blocks_in_target_range("Genome_2", "Genome_0", "Genome_0_seq", 0, 3000, seq=True, dup_mode=2, adjacencies=True)
    got = hal.format_block_results(blocks, dupes)
    for name in ("blockVizMmapTests.out", "blockVizHdf5Tests.out"):
        assert got == open(os.path.join(GOLD, "ref_blockviz", name)).read()


@pytest.mark.parametrize("seed", [2, 5])
def test_random_alignments_against_the_oracle(hal, oracle_bin, tmp_path, seed):
    """every genome pair (self-alignments included: they climb to the root), ranges of several sizes, with and without adjacencies,
    the three dupe modes, DNA, a coalescence limit, the reversed range of liftover mode; and all ranges of a pair in ONE call"""
    al, img = _rand_alignment(hal, tmp_path, seed, with_dna=True)
    n = al.num_genomes
    rng = np.random.default_rng(seed)
    lines = 0
    for q in range(n):
        for t in range(n):
            chrom, _, length = al.sequences(t)[0]
            if length == 0 or al.sequences(q)[0][2] == 0:
                continue
            qn, tn = al.genome_name(q), al.genome_name(t)
            ranges = []
            for size in (1, 40, 700, 4000):
                size = min(size, length)
                a = int(rng.integers(0, length - size + 1))
                ranges.append((a, a + size))
            ranges.append((0, 0))  # tEnd 0: to the end of the sequence
            variants = [dict(), dict(adjacencies=False), dict(dup_mode=1), dict(dup_mode=0, seq=True)]
            if (q + t) % 3 == 0:
                variants.append(dict(adjacencies=False, dup_mode=1, t_reversed=True))
            m = al.mrca(q, t)
            if (q + 2 * t) % 4 == 0 and al.genome_parent(m) >= 0:
                variants.append(dict(coalescence_limit=al.genome_name(al.genome_parent(m))))
            for kw in variants[: 2 + (q * n + t) % 5]:
                got = al.blocks_in_target_ranges(qn, tn, chrom, ranges, **kw)
                for (a, b), (blocks, dupes) in zip(ranges, got):
                    want = _oracle(oracle_bin, img, qn, tn, chrom, a, b, seq=kw.get("seq", False), dup_mode=kw.get("dup_mode", 2),
                                   adj=kw.get("adjacencies", True), limit=kw.get("coalescence_limit"), reversed_=kw.get("t_reversed", False))
                    text = hal.format_block_results(blocks, dupes)
                    assert text == want, (seed, qn, tn, a, b, kw)
                    lines += text.count("\n")
    assert lines > 5000


def test_multiseq_alignments_against_the_oracle(hal, oracle_bin, tmp_path):
    """the independent generator's alignments: several sequences per genome, irregular segments, insertions and deletions"""
    lines = 0
    for seed in (1, 6):
        img = str(tmp_path / ("ms%d.hgx" % seed))
        halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=6))
        al = hal.Alignment.open(img, device=0)
        n = al.num_genomes
        for q in range(n):
            for t in range(n):
                for chrom, _, length in al.sequences(t):
                    if length == 0:
                        continue
                    ranges = [(0, 0), (length // 3, min(length, length // 3 + 25)), (length - 1, length)]
                    got = al.blocks_in_target_ranges(al.genome_name(q), al.genome_name(t), chrom, ranges, seq=True)
                    for (a, b), (blocks, dupes) in zip(ranges, got):
                        text = hal.format_block_results(blocks, dupes)
                        assert text == _oracle(oracle_bin, img, al.genome_name(q), al.genome_name(t), chrom, a, b, seq=True), (seed, q, t, chrom, a, b)
                        lines += text.count("\n")
    assert lines > 1000


def test_argument_errors_carry_the_references_messages(hal, tmp_path):
    al, _ = _rand_alignment(hal, tmp_path, 2)
    name, _, length = al.sequences(0)[0]
    with pytest.raises(hal.HgxError, match="invalid query range"):
        al.blocks_in_target_range("Genome_2", "Genome_0", name, 50, 10)
    with pytest.raises(hal.HgxError, match="tReversed can only be set when mapBackAdjacencies is 0"):
        al.blocks_in_target_range("Genome_2", "Genome_0", name, 0, 10, t_reversed=True)
    with pytest.raises(hal.HgxError, match="outside of target sequence"):
        al.blocks_in_target_range("Genome_2", "Genome_0", name, 0, length + 1)
    with pytest.raises(hal.HgxError, match="not found in alignment"):
        al.blocks_in_target_range("Nobody", "Genome_0", name, 0, 10)
    with pytest.raises(hal.HgxError, match="Could not find coalescence limit"):
        al.blocks_in_target_range("Genome_2", "Genome_0", name, 0, 10, coalescence_limit="Nobody")
