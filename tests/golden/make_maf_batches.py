"""Makes tests/golden/maf_batches.bin: the device's batches (which columns are heads, the heads' rows: hgx_columns.hip,
columnsHeadRowsHost) of a fixed list of hal2maf exports over small alignments of tests/halfix.py's generator, recorded on a GPU box
by the profiling build of the library (make -C hal_amd/csrc hostprof-lib; HGX_MAF_DUMP).  tests/test_maf_replay.py plays them back
to hal2maf's host side (the block state machine and the rendering) on a machine without a GPU and holds the text against the oracle.
usage (GPU box): python tests/golden/make_maf_batches.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (seed, n_genomes, max_children, root_len), then per genome the exports below, in this order
ALIGNMENTS = [(3, 5, 2, 150), (8, 6, 1, 180), (21, 5, 3, 120), ("randgen", 0, 0, 0)]
EXPORTS = [({}, []), (dict(max_block_len=4, keep_empty_ref_blocks=True), ["--maxBlockLen", "4", "--keepEmptyRefBlocks"]),
           (dict(no_dupes=True, only_sequence_names=True), ["--noDupes", "--onlySequenceNames"]),
           (dict(unique=True), ["--unique"]),                       # which columns are walked / written comes from the device (heads 2, 3)
           (dict(unique=True, max_ref_gap=5), ["--unique", "--maxRefGap", "5"]),  # the column-by-column path: visit caches replayed on the host
           (dict(print_tree=True), ["--printTree"]),                # the same path with the block's tree
           (dict(print_tree=True, no_dupes=True, max_block_len=9), ["--printTree", "--noDupes", "--maxBlockLen", "9"]),
           (dict(max_ref_gap=9), ["--maxRefGap", "9"]),             # the iterator with its stack of inserted / deleted ranges
           (dict(max_ref_gap=1000, no_dupes=True, max_block_len=13), ["--maxRefGap", "1000", "--noDupes", "--maxBlockLen", "13"])]
# hal2maf --printTree has no tree for a column whose first base is an insertion in a genome with bottom segments (the reference
# dereferences a null iterator there, maf/impl/halMafBlock.cpp:281-287), nor where a base of the tree is not in the column (with
# --noDupes the paralogs the tree walks to were left out: the reference asserts, :180): the library and the oracle both say so
NO_TREE = ("has no parent in a genome with bottom segments", "no block entry continues at this base")


def cases(hal, device, tmp):
    """yields (image path, genome name, text of the library, oracle arguments) for every export of the list"""
    import halfix
    for seed, ng, mc, rl in ALIGNMENTS:
        img = os.path.join(tmp, "a%s.hgx" % seed)
        if seed == "randgen":  # the generator of the reference's own tests (halRandGen --preset small --seed 0: every top segment has a parent)
            opts = hal.RandOptions(mean_degree=1.3, max_branch_length=2.0, min_genomes=3, max_genomes=5, min_segment_length=5,
                                   max_segment_length=25, min_segments=8, max_segments=20, seed=4, with_dna=True)
            hal.Alignment.random(opts, device=-1).save(img)
        else:
            halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=ng, max_children=mc, root_len=rl))
        al = hal.Alignment.open(img, device=device)
        for g in range(al.num_genomes):
            if al.genome_length(g) == 0:
                continue
            for kw, args in EXPORTS:
                try:
                    text = al.maf_export(g, **kw)
                except hal.HgxError as e:
                    if not any(m in str(e) for m in NO_TREE):
                        raise
                    text = None
                yield img, al.genome_name(g), text, ["--refGenome", al.genome_name(g)] + args


# hal2maf --global over an alignment a randomised soak found (profiles/scripts/r03_features_soak.py): the last walk of a leaf's pass
# is abandoned under a paralogy cycle that still inserts its remaining members (halColumnIterator.cpp:653-680)
GLOBAL_CASE = (1431, 6, 2, 807)


def global_case(hal, device, tmp):
    import halfix
    seed, ng, mc, rl = GLOBAL_CASE
    img = os.path.join(tmp, "g%d.hgx" % seed)
    halfix.write_hgx(img, halfix.random_multiseq_alignment(seed, n_genomes=ng, max_children=mc, root_len=rl))
    return img, hal.Alignment.open(img, device=device).maf_export_global()


if __name__ == "__main__":
    # usage (GPU box): python tests/golden/make_maf_batches.py [global]   (one recording per process: the library opens its file once)
    import tempfile
    which = sys.argv[1] if len(sys.argv) > 1 else "exports"
    out = os.path.join(HERE, "maf_batches.bin" if which == "exports" else "maf_global_batches.bin")
    if os.path.exists(out):
        os.remove(out)
    os.environ["HGX_MAF_DUMP"] = out
    os.environ["HGX_LIB_PATH"] = os.path.join(ROOT, "hal_amd", "libhgx_hostprof.so")
    import hal_amd as hal
    with tempfile.TemporaryDirectory() as tmp:
        n = sum(1 for _ in cases(hal, 0, tmp)) if which == "exports" else len(global_case(hal, 0, tmp)[1])
    print("%s recorded (%d), %d bytes" % (which, n, os.path.getsize(out)))
