"""Golden vectors transcribed from the reference's liftover unit test
(liftover/tests/halLiftoverTests.cpp): the hand-built 5-genome alignment of setupSharedAlignment
(:15-252; inversions, an insertion, paralogy rings, incommensurate tilings) expressed as tables, and the
literal input/expected BED strings of testOneBranchLifts (:272-317) and testMultiBranchLifts (:319-343).
Data only."""
from halfix import NULL, simple_genome

T, F = True, False
DNA100 = "CAAAAGCTGCCTCGGCGTAGCCAGGTGTAAGCTGGTATTGTTCTTGTGCATCTGGGCACCATTCTCTTGTTCGTAAATAGGCGACGCTGTCTTTTGGCCG"
DNA70 = "ATGTGTATGCTTGGGTCAACTCTCTTTTCAGATCCGGGCGGTCGTCCGTAATTATGTGCCGAATCTCCAC"

def genomes():
    # ids: 0 root, 1 child1, 2 leaf1, 3 leaf2, 4 leaf3 (addLeafGenome order: child slots root=[child1,leaf1],
    # child1=[leaf2,leaf3])
    root = simple_genome("root", -1, [1, 2], 100, [], [
        (0, 20, [(0, T), (0, T)]),
        (20, 20, [(NULL, F), (2, T)]),
        (40, 20, [(2, F), (1, F)]),
        (60, 20, [(3, T), (NULL, F)]),
        (80, 20, [(NULL, F), (4, F)]),
    ], DNA100)
    child1 = simple_genome("child1", 0, [3, 4], 100, [
        (0, 20, 0, T, 4),
        (20, 20, NULL, F, NULL),
        (40, 20, 2, F, NULL),
        (60, 20, 3, T, NULL),
        (80, 20, 0, F, 0),
    ], [
        (0, 20, [(0, T), (NULL, F)]),
        (20, 10, [(NULL, F), (0, T)]),
        (30, 5, [(1, F), (NULL, F)]),
        (35, 15, [(NULL, F), (2, F)]),
        (50, 20, [(4, T), (1, T)]),
        (70, 20, [(3, F), (3, T)]),
        (90, 10, [(NULL, F), (4, F)]),
    ], DNA100)
    leaf1 = simple_genome("leaf1", 0, [], 100, [
        (0, 20, 0, T, NULL),
        (20, 20, 2, F, NULL),
        (40, 20, 1, T, NULL),
        (60, 20, NULL, F, NULL),
        (80, 20, 4, F, NULL),
    ], [], DNA100)
    leaf2 = simple_genome("leaf2", 1, [], 70, [
        (0, 20, 0, T, NULL),
        (20, 5, 2, F, 2),
        (25, 5, 2, F, 1),
        (30, 20, 5, F, NULL),
        (50, 20, 4, T, NULL),
    ], [], DNA70)
    leaf3 = simple_genome("leaf3", 1, [], 100, [
        (0, 10, 1, T, NULL),
        (10, 20, 4, T, NULL),
        (30, 15, 3, F, NULL),
        (45, 20, 5, T, NULL),
        (65, 10, 6, F, NULL),
        (75, 25, NULL, F, NULL),
    ], [], DNA100)
    return [root, child1, leaf1, leaf2, leaf3]

# (srcGenome, tgtGenome, input BED, expected BED) — BED6 cases of the reference test
CASES = [
    ("child1", "root",
     "Sequence\t0\t20\tPARALOGY1REV\t0\t+\n"
     "Sequence\t60\t80\tREV\t0\t+\n"
     "Sequence\t20\t40\tINSERTION\t0\t+\n"
     "Sequence\t80\t100\tPARALOGY2\t0\t+\n",
     "Sequence\t0\t20\tPARALOGY1REV\t0\t-\n"
     "Sequence\t60\t80\tREV\t0\t-\n"
     "Sequence\t0\t20\tPARALOGY2\t0\t+\n"),
    ("leaf1", "root",
     "Sequence\t0\t5\tNORMALREV\t0\t+\n"
     "Sequence\t10\t30\tOVERLAP\t0\t+\n"
     "Sequence\t50\t70\tOVERLAPINSERTION\t0\t+\n"
     "Sequence\t70\t100\tOVERLAPINSERTION2\t0\t+\n",
     "Sequence\t15\t20\tNORMALREV\t0\t-\n"
     "Sequence\t0\t10\tOVERLAP\t0\t-\n"
     "Sequence\t40\t50\tOVERLAP\t0\t+\n"
     "Sequence\t20\t30\tOVERLAPINSERTION\t0\t-\n"
     "Sequence\t80\t100\tOVERLAPINSERTION2\t0\t+\n"),
    ("root", "child1",
     "Sequence\t0\t10\tPARALOGY\t0\t+\n"
     "Sequence\t30\t50\tOVERLAPINSERTION\t0\t+\n",
     "Sequence\t10\t20\tPARALOGY\t0\t-\n"
     "Sequence\t80\t90\tPARALOGY\t0\t+\n"
     "Sequence\t40\t50\tOVERLAPINSERTION\t0\t+\n"),
    ("leaf2", "leaf3",
     "Sequence\t30\t35\tREV\t0\t+\n"
     "Sequence\t40\t60\tOVERLAP\t0\t+\n",
     "Sequence\t60\t65\tREV\t0\t-\n"
     "Sequence\t45\t55\tOVERLAP\t0\t-\n"
     "Sequence\t10\t20\tOVERLAP\t0\t+\n"),
    ("root", "leaf2",
     "Sequence\t0\t20\tBLOCK_A\t0\t+\n"
     "Sequence\t30\t50\tBLOCK_B\t0\t+\n",
     "Sequence\t0\t20\tBLOCK_A\t0\t+\n"
     "Sequence\t40\t50\tBLOCK_A\t0\t+\n"),
]

# leaf3 -> leaf1 (up 2, down 1) with BED12 blocks: liftover/tests/halLiftoverTests.cpp:345-373
CASE3_BED12 = ("Sequence\t0\t10\tSEGMENT_0\t0\t+\t0\t10\t128,0,0\t1\t10\t0,\n"
               "Sequence\t10\t30\tSEGMENT_1\t0\t+\t10\t30\t128,0,0\t1\t20\t0,\n"
               "Sequence\t30\t45\tSEGMENT_2\t0\t+\t30\t45\t128,0,0\t1\t15\t0,\n"
               "Sequence\t45\t65\tSEGMENT_3\t0\t+\t45\t65\t128,0,0\t1\t20\t0,\n"
               "Sequence\t65\t75\tSEGMENT_4\t0\t+\t65\t75\t128,0,0\t1\t10\t0,\n"
               "Sequence\t75\t100\tSEGMENT_5\t0\t+\t75\t100\t128,0,0\t1\t25\t0,\n")
# (srcGenome, tgtGenome, input, expected, outPSL, outPSLWithName)
CASES12 = [
    ("leaf3", "leaf1", CASE3_BED12,
     "Sequence\t30\t40\tSEGMENT_1\t0\t-\t30\t40\t128,0,0\t1\t10\t0\n"
     "Sequence\t20\t30\tSEGMENT_2\t0\t+\t20\t30\t128,0,0\t1\t10\t0\n"
     "Sequence\t10\t20\tSEGMENT_3\t0\t+\t10\t20\t128,0,0\t1\t10\t0\n"
     "Sequence\t0\t10\tSEGMENT_4\t0\t-\t0\t10\t128,0,0\t1\t10\t0\n", False, False),
    ("leaf3", "leaf1", CASE3_BED12,
     "2\t8\t0\t0\t0\t0\t0\t0\t+-\tSequence\t100\t20\t30\tSequence\t100\t30\t40\t1\t10,\t20,\t60,\n"
     "2\t8\t0\t0\t0\t0\t0\t0\t++\tSequence\t100\t35\t45\tSequence\t100\t20\t30\t1\t10,\t35,\t20,\n"
     "3\t7\t0\t0\t0\t0\t0\t0\t++\tSequence\t100\t45\t55\tSequence\t100\t10\t20\t1\t10,\t45,\t10,\n"
     "3\t7\t0\t0\t0\t0\t0\t0\t+-\tSequence\t100\t65\t75\tSequence\t100\t0\t10\t1\t10,\t65,\t90,\n", True, False),
    ("leaf3", "leaf1", CASE3_BED12,
     "SEGMENT_1\t2\t8\t0\t0\t0\t0\t0\t0\t+-\tSequence\t100\t20\t30\tSequence\t100\t30\t40\t1\t10,\t20,\t60,\n"
     "SEGMENT_2\t2\t8\t0\t0\t0\t0\t0\t0\t++\tSequence\t100\t35\t45\tSequence\t100\t20\t30\t1\t10,\t35,\t20,\n"
     "SEGMENT_3\t3\t7\t0\t0\t0\t0\t0\t0\t++\tSequence\t100\t45\t55\tSequence\t100\t10\t20\t1\t10,\t45,\t10,\n"
     "SEGMENT_4\t3\t7\t0\t0\t0\t0\t0\t0\t+-\tSequence\t100\t65\t75\tSequence\t100\t0\t10\t1\t10,\t65,\t90,\n", True, True),
]


def extra_paralogs_genomes():
    """The alignment of MappedSegmentMapExtraParalogsTest (api/tests/halMappedSegmentTest.cpp:478-566): every segment of
    grandChild1 coalesces with the first segment of grandChild2, but only above the MRCA (in `root`)."""
    from halfix import simple_genome
    root = simple_genome("root", -1, [1], 3, [], [(0, 3, [(0, False)])], dna="CCC")
    parent = simple_genome("parent", 0, [2, 3], 9, [(0, 3, 0, False, 1), (3, 3, 0, False, 2), (6, 3, 0, False, 0)],
                           [(0, 3, [(0, True), (0, False)]), (3, 3, [(1, True), (-1, True)]), (6, 3, [(2, True), (-1, False)])],
                           dna="CCCTACGTG")
    gc1 = simple_genome("grandChild1", 1, [], 9, [(0, 3, 0, True, -1), (3, 3, 1, True, -1), (6, 3, 2, True, -1)], [], dna="CCCTACGTG")
    gc2 = simple_genome("grandChild2", 1, [], 9, [(0, 3, 0, False, -1), (3, 3, -1, True, -1), (6, 3, -1, False, -1)], [], dna="CCCTACGTG")
    return [root, parent, gc1, gc2]


# :568-611: mapping grandChild2's first top segment to grandChild1: by default only the homology inside the MRCA
# (target start position 2, length 3, reversed = forward range [0, 3)); with the root as coalescence limit all three
# paralogs (start positions 2, 5, 8, reversed).  Lines of `hal_oracle blocks` / hgx_block_map: target sequence, forward
# target range, forward source start, source strand, target strand.
EXTRA_PARALOGS_DEFAULT = "Sequence\t0\t3\t0\t+\t-\n"
EXTRA_PARALOGS_ROOT_LIMIT = "Sequence\t0\t3\t0\t+\t-\nSequence\t3\t6\t0\t+\t-\nSequence\t6\t9\t0\t+\t-\n"
