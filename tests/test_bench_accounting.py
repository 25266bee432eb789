"""bench.py's byte accounting on the host: the per-kernel algorithmic bytes of the single-pass kernels, the
tree sweeps' own bytes on a small alignment, and the identity of the device sources that profiles/pmc_traffic.json is tied to."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_single_pass_kernel_bytes():
    st = {"queries": 1000, "records": 3000}
    kt = {"k_lift_classify": {"launches": 2, "top_derefs": 40},    # 40 unmerged records clipped by general intervals
          "k_lift_merged": {"launches": 2, "top_derefs": 5800},    # merged records that overlap their interval, both steps
          "k_lift_totals": {"launches": 2}}
    b = bench.plan_kernel_bytes(kt, st, steps=2)
    assert b["k_lift_merged"] == (16.0 * 5800 + 37.0 * 1000 * 2 + 40.0 * 3000 * 2) / 2
    assert b["k_lift_classify"] == (32.0 * 1000 * 2 + 16.0 * (5800 + 40)) / 2
    assert b["k_lift_totals"] == 0.0


def test_sweeps_own_bytes_on_a_three_genome_tree(hal):
    opts = hal.RandOptions(mean_degree=2.0, max_branch_length=1.0, min_genomes=3, max_genomes=3, min_segment_length=10, max_segment_length=20,
                           min_segments=50, max_segments=60, seed=1, with_dna=False)
    al = hal.Alignment.random(opts, device=-1)
    assert al.num_genomes == 3
    root = [g for g in range(3) if al.genome_parent(g) < 0][0]
    kids = al.genome_children(root)
    ref = kids[0]
    word = 1  # three genomes: an 8-bit genome set per base
    want = word * al.genome_length(root) + al.num_bottom_segments(root) * (8.0 + 4.0 * len(kids)) + sum(16.0 * al.num_top_segments(c) for c in kids)
    if len(kids) == 2:  # (leaves carry constants: no child track is read)
        # (down: TopRec + the parent's BotRec, the depth written as a byte, the root's track read; out: the byte read, int32 written)
        want += 24.0 * al.num_top_segments(ref) + 1.0 * al.genome_length(ref) + word * al.genome_length(ref) + 5.0 * al.genome_length(ref)
        assert bench.sweep_design_bytes(al, ref) == want


def test_device_source_identity_is_stable():
    a, b = bench.kernel_sources_sha16(), bench.kernel_sources_sha16()
    assert a == b and len(a) == 16
