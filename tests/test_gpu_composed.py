"""The composed tables (large plans) against the oracle — the merged table of the whole path with the single-pass kernels
("merged", the default: k_lift_classify / k_lift_merged), the unmerged table of the whole path with the multi-kernel
finishing step ("through", HGX_MERGED=0: k_locate_through) and the up table source -> MRCA ("up", HGX_COMPOSED_THROUGH=0):
forced with HGX_COMPOSED_UP=1 on small batches, every genome pair, both strands and '.', dupes on and off, BED12 / PSL,
the coalescence limit, real data; and at scale against the walk kernels (HGX_COMPOSED_UP=0)."""
import os

import numpy as np
import pytest

import halfix
from util import oracle_liftover, random_bed
from test_gpu_liftover import _rand_alignment
from test_gpu_multiseq import _bed

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True, params=["merged", "through", "up"])
def _forced(monkeypatch, request):
    monkeypatch.setenv("HGX_COMPOSED_UP", "1")
    monkeypatch.setenv("HGX_COMPOSED_THROUGH", "0" if request.param == "up" else "1")
    monkeypatch.setenv("HGX_MERGED", "1" if request.param == "merged" else "0")
    return request.param


TABLE_KERNEL = {"merged": "k_lift_merged", "through": "k_locate_through", "up": "k_locate_composed"}
TABLE_KIND = {"merged": 3, "through": 2, "up": 1}


@pytest.mark.parametrize("seed", [2, 5, 9])
def test_randgen_all_pairs(hal, oracle_bin, tmp_path, seed):
    al, img = _rand_alignment(hal, tmp_path, seed)
    n = al.num_genomes
    lines = 0
    for s in range(n):
        name, _, length = al.sequences(s)[0]
        if length == 0:
            continue
        for t in range(n):
            bed = random_bed(name, length, 120, 1, 400, seed * 100 + s * n + t, strands="+-.")
            nd = (s + t) % 3 == 0
            got = hal.liftover_convert(al, s, bed, t, traverse_dupes=not nd)
            assert got == oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, no_dupes=nd), \
                (al.genome_name(s), al.genome_name(t), nd)
            lines += got.count("\n")
            m = al.mrca(s, t)
            if (s + 2 * t) % 5 == 0 and al.genome_parent(m) >= 0:
                lim = al.genome_parent(m)
                assert hal.liftover_convert(al, s, bed, t, coalescence_limit=lim) == \
                    oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path, coalescence_limit=al.genome_name(lim))
    assert lines > 2000


@pytest.mark.parametrize("seed", [0, 3])
def test_multiseq_bed12_psl(hal, oracle_bin, tmp_path, seed):
    img = str(tmp_path / "ms.hgx")
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=7))
    al = hal.Alignment.open(img, device=0)
    n = al.num_genomes
    for s in range(n):
        for t in range(n):
            bed = _bed(al, s, 50, seed * 100 + s * n + t)
            assert hal.liftover_convert(al, s, bed, t) == oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bed, tmp_path)
            if (s + t) % 4 == 0:
                bedp = bed.replace("\t.\n", "\t+\n")
                assert hal.liftover_convert(al, s, bedp, t, out_psl=True) == \
                    oracle_liftover(oracle_bin, img, al.genome_name(s), al.genome_name(t), bedp, tmp_path, psl=True)


def test_real_data(hal, oracle_bin, tmp_path):
    try:
        al = hal.Alignment.open(os.path.join(GOLD, "ref_hdf5", "mr.hal"), device=0)
    except hal.HgxError as e:
        if "HDF5 C library" in str(e):
            pytest.skip("libhdf5 not loadable here")
        raise
    img = str(tmp_path / "mr.hgx")
    al.save(img)
    for src, tgt in (("simMouse_chr6", "simRat_chr6"), ("simRat_chr6", "mr"), ("simRat_chr6", "simMouse_chr6")):
        s, t = al.genome_id(src), al.genome_id(tgt)
        bed = _bed(al, s, 2000, 3 + s + t)
        name, _, length = al.sequences(s)[0]
        rng = np.random.default_rng(8)
        for _ in range(60):
            ln = int(rng.integers(1000, 30000))
            st = int(rng.integers(0, length - ln))
            bed += "%s\t%d\t%d\tlong\t0\t%s\n" % (name, st, st + ln, "+-"[int(rng.integers(0, 2))])
        assert hal.liftover_convert(al, s, bed, t) == oracle_liftover(oracle_bin, img, src, tgt, bed, tmp_path), (src, tgt)


def test_at_scale_against_the_walk_kernels(hal, monkeypatch, _forced):
    """200 k intervals on ~10 Mb genomes: composed table vs level walk, record for record."""
    import torch
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=50,
                           max_segment_length=200, min_segments=70000, max_segments=140000, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_8")
    _, ss, length = al.sequences(src)[0]
    n = 200000
    g = torch.Generator().manual_seed(4)
    starts = torch.randint(0, length - 1100, (n,), generator=g)
    lens = torch.randint(50, 1000, (n,), generator=g)
    strand = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8)
    gs, ge, st = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda(), strand.cuda()
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HGX_COMPOSED_UP", mode)
        plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
        ptr, nrec = plan.run(gs, ge, st)
        out[mode] = plan.records_to_tensor(ptr, nrec).cpu()
        kt = plan.kernel_times()
        assert (TABLE_KERNEL[_forced] in kt) == (mode == "1") and ("k_up_chain" in kt) == (mode == "0")
        if mode == "1":  # the table of the whole path replaces the down hop and the grouping scatter as well
            assert ("k_down_ring" in kt) == (_forced == "up") and ("k_scatter" in kt) == (_forced == "up")
            assert plan.stats()["composed_kind"] == TABLE_KIND[_forced]
            if _forced == "merged":  # one kernel writes the records: no register finishing step, no compaction
                assert "k_finish_fast" not in kt and "k_compact_records" not in kt
    assert out["1"].shape[0] > n and torch.equal(out["1"], out["0"])


def test_a_walking_plan_switches_to_its_table(hal, monkeypatch, _forced):
    """Without HGX_COMPOSED_UP a plan walks until its intervals reach a multiple of the source's segments (a quarter by
    default, four times here), then builds and uses the table; the records before and after the switch are the same."""
    import torch
    monkeypatch.delenv("HGX_COMPOSED_UP")
    monkeypatch.setenv("HGX_COMPOSED_AFTER", "4")
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=20,
                           max_segment_length=80, min_segments=2000, max_segments=4000, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_8")
    _, ss, length = al.sequences(src)[0]
    n = 3000
    g = torch.Generator().manual_seed(11)
    starts = torch.randint(0, length - 400, (n,), generator=g)
    lens = torch.randint(1, 300, (n,), generator=g)
    gs, ge = (starts + ss).cuda(), (starts + lens - 1 + ss).cuda()
    st = torch.where(torch.rand(n, generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8).cuda()
    plan = hal.LiftoverPlan(al, src, tgt, max_queries=n)
    table_kernel = TABLE_KERNEL[_forced]
    first, kinds = None, []
    for _ in range(12):
        ptr, nrec = plan.run(gs, ge, st)
        recs = plan.records_to_tensor(ptr, nrec).cpu()
        if first is None:
            first = recs
        assert torch.equal(recs, first)
        kinds.append(table_kernel in plan.kernel_times())
    assert kinds[0] is False and kinds[-1] is True and sorted(kinds) == kinds  # walks first, one switch, the table afterwards
    assert plan.stats()["composed_kind"] == TABLE_KIND[_forced]


def test_intervals_at_and_over_the_edges(hal, monkeypatch, _forced):
    """What the device entry point is handed unchecked: intervals that end behind the genome, begin in front of it, are empty,
    are one base, cover the whole genome, sit on the first and the last base — table against walk, record for record."""
    import torch
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=20,
                           max_segment_length=80, min_segments=2000, max_segments=4000, seed=2, with_dna=False)
    al = hal.Alignment.random(opts, device=0)
    src, tgt = al.genome_id("Genome_9"), al.genome_id("Genome_8")
    _, ss, length = al.sequences(src)[0]
    edge = [(0, 0), (0, 99), (length - 1, length - 1), (length - 100, length - 1), (length - 100, length + 50), (length - 1, 2 ** 40),
            (length, length + 10), (-5, 20), (50, 40), (0, length - 1), (0, 2 ** 31), (17, 17), (length // 2, length // 2 + 20000)]
    g = torch.Generator().manual_seed(5)
    n = 700
    starts = torch.cat([torch.tensor([a for a, _ in edge]), torch.randint(0, length - 400, (n,), generator=g)])
    ends = torch.cat([torch.tensor([b for _, b in edge]), torch.zeros(n, dtype=torch.int64)])
    ends[len(edge):] = starts[len(edge):] + torch.randint(0, 300, (n,), generator=g)
    st = torch.where(torch.rand(starts.numel(), generator=g) < 0.5, ord("+"), ord("-")).to(torch.uint8).cuda()
    gs, ge = (starts + ss).cuda(), (ends + ss).cuda()
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HGX_COMPOSED_UP", mode)
        plan = hal.LiftoverPlan(al, src, tgt, max_queries=int(starts.numel()))
        ptr, nrec = plan.run(gs, ge, st)
        out[mode] = plan.records_to_tensor(ptr, nrec).cpu()
    assert out["1"].shape[0] > n and torch.equal(out["1"], out["0"])
