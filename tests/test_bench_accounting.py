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


def test_kernel_code_identity_per_kernel():
    """the PMC traffic of a kernel is tied to the hash of that kernel's machine code in libhgx.so (bench.kernel_code_sha16s): every
    timed kernel has one, two readings agree, and a recorded value voids exactly its own kernel"""
    import json
    import tempfile
    ids = bench.kernel_code_sha16s()
    for k in ("k_lift_classify", "k_lift_merged", "k_lift_totals", "k_up_chain", "k_sweep_up", "k_sweep_down", "k_finish_lds"):
        assert len(ids.get(k, "")) == 16, k
    bench._KERNEL_CODE.clear()
    assert bench.kernel_code_sha16s() == ids
    real = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    saved = open(real).read() if os.path.exists(real) else None
    try:
        with open(real, "w") as f:
            json.dump({"libhgx_sha16": "x", "kernel_sources_sha16": "y", "source": "test",
                       "kernel_code_sha16": {"k_lift_classify": ids["k_lift_classify"], "k_lift_merged": "0" * 16},
                       "rotating": {"k_lift_classify": 1.0e8, "k_lift_merged": 2.0e8}}, f)
        assert bench.pmc_traffic("k_lift_classify", "rotating")[0] == 1.0e8
        assert bench.pmc_traffic("k_lift_merged", "rotating")[0] is None
    finally:
        if saved is None:
            os.unlink(real)
        else:
            open(real, "w").write(saved)


def test_cpu_column_baselines_and_their_parity_gates(hal, tmp_path):
    """bench.py's CPU figures beside the column legs: the oracle's hal2maf / halAlignmentDepth loops over a genome's first columns,
    timed, and the gates that hold a timed export's text against them — here the whole genome's text (the oracle's own) stands in
    for the GPU's: the wig of a slice is the beginning of the genome's wig, the MAF of a slice, up to its last block, the
    beginning of the genome's MAF; a text that differs anywhere before fails the gate."""
    opts = hal.RandOptions(mean_degree=1.5, max_branch_length=3.0, min_genomes=2, max_genomes=10, min_segment_length=10, max_segment_length=60,
                           min_segments=150, max_segments=300, seed=2, with_dna=True)
    al = hal.Alignment.random(opts, device=-1)
    img = str(tmp_path / "a.hgx")
    al.save(img)
    g = al.genome_id("Genome_9")
    seq, _, n = al.sequences(g)[0]
    for kind, extra in (("depth", ()), ("maf", ("--noAncestors",))):
        whole, text_whole = bench.cpu_columns_baseline(img, kind, "Genome_9", seq, n, str(tmp_path), "w", extra=extra)
        part, text_part = bench.cpu_columns_baseline(img, kind, "Genome_9", seq, n // 3, str(tmp_path), "p", all_cores_total=n // 2, extra=extra)
        assert whole["value"] > 0 and part["cores"] == 1 and part["kind"] == "port"
        assert part["all_cores"]["cores"] == (os.cpu_count() or 1) and part["all_cores"]["value"] > 0
        if kind == "depth":
            assert text_whole[:len(text_part)] == text_part and text_part.count(b"\n") == n // 3 + 1
        else:
            assert bench.maf_prefix_matches(text_part, text_whole[:len(text_part) + 100])
            broken = bytearray(text_whole)
            at = len(text_part) // 2
            broken[at] = ord("X") if broken[at] != ord("X") else ord("Y")
            assert not bench.maf_prefix_matches(text_part, bytes(broken))
            assert not bench.maf_prefix_matches(text_part, text_whole[:len(text_part) // 2])
