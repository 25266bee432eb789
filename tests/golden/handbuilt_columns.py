"""The hand-built alignments of the reference's ColumnIterator unit tests (api/tests/halColumnIteratorTest.cpp) re-expressed
as tables, and the facts those tests assert, as predicates over the rows of every column.  Data only.

  depth()  ColumnIteratorDepthTest  :55-156   grandpa - dad - (son1, son2), ten 10-base segments, everything collinear
  dup()    ColumnIteratorDupTest    :158-280  dad - (son1, son2); son1 is ten copies of dad's segment 0, son2 has segments 4
                                              and 8 both derived from dad's segment 4
  inv()    ColumnIteratorInvTest    :282-457  grandpa - dad - son1 with inversions on one and on both branches
  gap()            ColumnIteratorGapTest          :459-540  grandpa - dad; dad lacks grandpa's middle segment (a deletion); iterated from
                                                            dad with maxInsertLength 1000 the deleted bases come as columns of their own
  multi_gap()      ColumnIteratorMultiGapTest     :542-732  adam - grandpa - dad, a deletion on either branch (nested stack entries)
  multi_gap_inv()  ColumnIteratorMultiGapInvTest  :734-933  the same with adam's second segment inverted in grandpa
"""
from halfix import NULL, fix_parse_info

N, L = 10, 10


def _genome(name, parent, children, top, bot, dna):
    """top: list of (parentIdx, parentRev, nextParalogy) or None; bot: list of per-child (idx, rev) lists or None"""
    g = {"name": name, "parent": parent, "children": children}
    nt, nb = (N if top is not None else 0), (N if bot is not None else 0)
    g["tStart"] = [i * L for i in range(nt)] + [N * L]
    g["tParent"] = [t[0] for t in top] if top else []
    g["tParentRev"] = [1 if t[1] else 0 for t in top] if top else []
    g["tParalogy"] = [t[2] for t in top] if top else []
    g["bStart"] = [i * L for i in range(nb)] + [N * L]
    g["bChild"] = [[bot[i][k][0] for i in range(nb)] for k in range(len(children))] if bot else [[] for _ in children]
    g["bChildRev"] = [[1 if bot[i][k][1] else 0 for i in range(nb)] for k in range(len(children))] if bot else [[] for _ in children]
    g["seqs"] = [("seq", 0, N * L, 0, nt, 0, nb)]
    g["dna"] = dna
    fix_parse_info(g)
    return g


def depth():
    plain_top = [(i, False, NULL) for i in range(N)]
    return [_genome("grandpa", -1, [1], None, [[(i, False)] for i in range(N)], "T" * 100),
            _genome("dad", 0, [2, 3], plain_top, [[(i, False), (i, False)] for i in range(N)], "G" * 100),
            _genome("son1", 1, [], plain_top, None, "A" * 100),
            _genome("son2", 1, [], plain_top, None, "C" * 100)]


def check_depth(ref_name, col, rows):
    """:126-141: four sequences per column, one base each, all at array index == column"""
    assert sorted(g for g, _, _ in rows) == ["dad", "grandpa", "son1", "son2"], (ref_name, col, rows)
    assert all(p == col for _, p, _ in rows), (ref_name, col, rows)


def dup():
    son1 = [(0, False, (i + 1) % N) for i in range(N)]                       # :207-219: all from dad's 0, one ring
    son2 = [(i, False, NULL) for i in range(N)]
    son2[4] = (4, False, 8)                                                  # :221-229
    son2[8] = (4, False, 4)
    dad_bot = []
    for i in range(N):
        c0 = (i, False) if i == 0 else (NULL, False)                         # :216-218 child 0 only from segment 0
        c1 = (NULL, False) if i == 8 else (i, False)                         # :228-229
        dad_bot.append([c0, c1])
    return [_genome("dad", -1, [1, 2], None, dad_bot, "G" * 100), _genome("son1", 0, [], son1, None, "t" * 100),
            _genome("son2", 0, [], son2, None, "c" * 100)]


def check_dup(ref_name, col, rows):
    """:232-268"""
    n1 = sum(1 for g, _, _ in rows if g == "son1")
    n2 = sum(1 for g, _, _ in rows if g == "son2")
    if ref_name != "son1":
        assert n1 == (10 if col < 10 else 0), (ref_name, col, rows)          # every son1 segment aligns to the first segment
    if ref_name == "dad":
        assert n2 == (2 if 40 <= col < 50 else 0 if 80 <= col < 90 else 1), (ref_name, col, rows)


def inv():
    gdna = "".join("AGTC"[i % 4] for i in range(100))                        # :338-366
    ddna = "".join("CAGT"[i % 4] for i in range(100))
    sdna = "".join("TCAG"[i % 4] for i in range(100))
    son1 = [(i, i in (0, 1), NULL) for i in range(N)]                        # :372-382 child-dad edge inverted in segments 0, 1
    dad_top = [(i, i == 1, NULL) for i in range(N)]                          # :383-386 dad-grandpa edge inverted in segment 1
    dad_bot = [[(i, i in (0, 1))] for i in range(N)]
    gra_bot = [[(i, i == 1)] for i in range(N)]
    return [_genome("grandpa", -1, [1], None, gra_bot, gdna), _genome("dad", 0, [2], dad_top, dad_bot, ddna),
            _genome("son1", 1, [], son1, None, sdna)]


def check_inv(ref_name, col, rows):
    """:389-450 (reference = son1)"""
    assert ref_name == "son1"
    by = {g: (p, r) for g, p, r in rows}
    assert sorted(by) == ["dad", "grandpa", "son1"] and len(rows) == 3, (col, rows)
    if col < 10:
        assert by["son1"] == (col, False) and by["dad"] == (9 - col, True) and by["grandpa"] == (9 - col, True), (col, rows)
    elif col < 20:
        assert by["son1"] == (col, False) and by["dad"] == (29 - col, True) and by["grandpa"] == (col, False), (col, rows)
    else:
        assert by["son1"][0] == by["dad"][0] == by["grandpa"][0] == col, (col, rows)


def _small(name, parent, children, length, tops, bots, dna, seq):
    """4-base segments: tops = list of (parentIdx, parentRev), bots = list of per-child (idx, rev)"""
    g = {"name": name, "parent": parent, "children": children}
    g["tStart"] = [4 * i for i in range(len(tops))] + [length]
    g["tParent"] = [t[0] for t in tops]
    g["tParentRev"] = [1 if t[1] else 0 for t in tops]
    g["tParalogy"] = [NULL] * len(tops)
    g["bStart"] = [4 * i for i in range(len(bots))] + [length]
    g["bChild"] = [[b[k][0] for b in bots] for k in range(len(children))]
    g["bChildRev"] = [[1 if b[k][1] else 0 for b in bots] for k in range(len(children))]
    g["seqs"] = [(seq, 0, length, 0, len(tops), 0, len(bots))]
    g["dna"] = dna
    fix_parse_info(g)
    return g


def gap():
    # :476-507
    return [_small("grandpa", -1, [1], 12, [], [[(0, False)], [(NULL, False)], [(1, False)]], "ACGTAAAAGGGG", "gseq"),
            _small("dad", 0, [], 8, [(0, False), (2, False)], [], "ACGTGGGG", "dseq")]


def check_gap(i, rows):
    """:519-538: column i of the iteration from dad with maxInsertLength 1000 (12 columns)"""
    by = {}
    for g, p, _ in rows:
        by.setdefault(g, []).append(p)
    if i < 4 or i >= 8:
        assert by.get("dad") == [i if i < 4 else i - 4], (i, rows)
    assert by.get("grandpa") == [i], (i, rows)


def _multi_gap(inverted):
    # :566-640 / :758-838
    adam = _small("adam", -1, [1], 16, [], [[(0, False)], [(1, inverted)], [(NULL, False)], [(2, False)]], "ACGTAAAATTTTGGGG", "aseq")
    grandpa = _small("grandpa", 0, [2], 12, [(0, False), (1, inverted), (3, False)], [[(0, False)], [(NULL, False)], [(1, False)]],
                     "ACGTAAAAGGGG", "gseq")
    dad = _small("dad", 1, [], 8, [(0, False), (2, False)], [], "ACGTGGGG", "dseq")
    return [adam, grandpa, dad]


def multi_gap():
    return _multi_gap(False)


def multi_gap_inv():
    return _multi_gap(True)


def _by(rows):
    by = {}
    for g, p, _ in rows:
        by.setdefault(g, []).append(p)
    return by


def check_multi_gap(i, rows):
    """:660-729 (16 columns)"""
    by = _by(rows)
    if i < 4:
        assert by == {"adam": [i], "grandpa": [i], "dad": [i]}, (i, rows)
    elif i < 8:
        assert by == {"adam": [i], "grandpa": [i]}, (i, rows)
    elif i < 12:
        assert by == {"adam": [i]}, (i, rows)
    else:
        assert by == {"adam": [i], "grandpa": [i - 4], "dad": [i - 8]}, (i, rows)


def check_multi_gap_inv(i, rows):
    """:854-930 (16 columns; the assertion on adam's position in columns 8..11 is disabled in the reference)"""
    by = _by(rows)
    if i < 4:
        assert by == {"adam": [i], "grandpa": [i], "dad": [i]}, (i, rows)
    elif i < 8:
        assert by == {"adam": [11 - i], "grandpa": [i]}, (i, rows)
    elif i < 12:
        assert sorted(by) == ["adam"] and len(by["adam"]) == 1, (i, rows)
    else:
        assert by == {"adam": [i], "grandpa": [i - 4], "dad": [i - 8]}, (i, rows)


# (name, alignment, check(column number in iteration order, rows), reference genome, number of columns) with maxInsertLength 1000
GAP_CASES = [("gap", gap, check_gap, "dad", 12), ("multi_gap", multi_gap, check_multi_gap, "dad", 16),
             ("multi_gap_inv", multi_gap_inv, check_multi_gap_inv, "dad", 16)]

CASES = [("depth", depth, check_depth, ["grandpa", "dad", "son1", "son2"]), ("dup", dup, check_dup, ["dad", "son1", "son2"]),
         ("inv", inv, check_inv, ["son1"])]
