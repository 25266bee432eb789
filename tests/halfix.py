"""Test-side helpers: build alignment images from plain Python data and write them as HGX flat
images (format: DESIGN.md section 3).  Independent of hal_amd's C++ writer on purpose."""
import struct

NULL = -1

def _pad(n):
    return b"\0" * ((8 - n % 8) % 8)

def _str(s):
    b = s.encode()
    return struct.pack("<q", len(b)) + b + _pad(len(b))

def _a64(v):
    return struct.pack("<%dq" % len(v), *v)

def _a8(v):
    b = bytes(v)
    return b + _pad(len(b))

_PACK = {c: i for i, c in enumerate("acgtn")}
_PACK.update({c.upper(): i + 8 for c, i in list(_PACK.items())})

def pack_dna(s):
    out = bytearray((len(s) + 1) // 2)
    for i, c in enumerate(s):
        code = _PACK.get(c, 4)
        out[i >> 1] |= code if (i & 1) else (code << 4)
    return bytes(out)

def fix_parse_info(g):
    """Parse indices = index of the segment of the other tiling that contains this segment's start
    (what Genome::fixParseInfo, api/impl/halGenome.cpp:263-307, computes)."""
    ts, bs = g["tStart"], g["bStart"]
    nt, nb = len(ts) - 1, len(bs) - 1
    g["tBotParse"] = [NULL] * nt
    g["bTopParse"] = [NULL] * nb
    if nt == 0 or nb == 0:
        return
    j = 0
    for i in range(nt):
        while bs[j + 1] <= ts[i]:
            j += 1
        g["tBotParse"][i] = j
    j = 0
    for i in range(nb):
        while ts[j + 1] <= bs[i]:
            j += 1
        g["bTopParse"][i] = j

def newick(genomes, root=None):
    if root is None:
        root = [i for i, g in enumerate(genomes) if g["parent"] < 0][0]
    def rec(i):
        g = genomes[i]
        s = ""
        if g["children"]:
            s += "(" + ",".join(rec(c) for c in g["children"]) + ")"
        s += g["name"]
        if g["parent"] >= 0:
            s += ":%g" % g.get("branch", 1)
        return s
    return rec(root) + ";"

def write_hgx(path, genomes):
    """genomes: list of dicts with name, parent, children, seqs [(name,start,len,topStart,numTop,botStart,numBot)],
    tStart(+sentinel), tParent, tParalogy, tBotParse, tParentRev, bStart(+sentinel), bTopParse,
    bChild [slot][seg], bChildRev [slot][seg], dna (string)."""
    out = [b"HGXIMG01", struct.pack("<q", len(genomes)), _str(newick(genomes))]
    for g in genomes:
        nt, nb = len(g["tStart"]) - 1, len(g["bStart"]) - 1
        total = g["tStart"][-1] if nt > 0 else g["bStart"][-1]
        out += [_str(g["name"]), struct.pack("<q", g["parent"]), struct.pack("<q", len(g["children"]))]
        out += [struct.pack("<q", c) for c in g["children"]]
        out += [struct.pack("<4q", total, len(g["seqs"]), nt, nb)]
        for s in g["seqs"]:
            out += [_str(s[0]), struct.pack("<6q", *s[1:])]
        out += [_a64(g["tStart"]), _a64(g["tParent"]), _a64(g["tParalogy"]), _a64(g["tBotParse"]), _a8(g["tParentRev"])]
        out += [_a64(g["bStart"]), _a64(g["bTopParse"])]
        for k in range(len(g["children"])):
            out += [_a64(g["bChild"][k]), _a8(g["bChildRev"][k])]
        dna = pack_dna(g.get("dna", "N" * total))
        out += [struct.pack("<q", len(dna)), _a8(dna)]
    with open(path, "wb") as f:
        f.write(b"".join(out))

def simple_genome(name, parent, children, length, tops, bots, dna=None, seqname="Sequence"):
    """tops: list of (start, len, parentIdx, parentRev, nextParalogy); bots: list of
    (start, len, [(childIdx, childRev), ...])."""
    g = {"name": name, "parent": parent, "children": children}
    g["tStart"] = [t[0] for t in tops] + [length]
    g["tParent"] = [t[2] for t in tops]
    g["tParentRev"] = [1 if t[3] else 0 for t in tops]
    g["tParalogy"] = [t[4] for t in tops]
    g["bStart"] = [b[0] for b in bots] + [length]
    g["bChild"] = [[b[2][k][0] for b in bots] for k in range(len(children))]
    g["bChildRev"] = [[1 if b[2][k][1] else 0 for b in bots] for k in range(len(children))]
    for i, t in enumerate(tops):
        assert g["tStart"][i + 1] - t[0] == t[1], (name, "top", i)
    for i, b in enumerate(bots):
        assert g["bStart"][i + 1] - b[0] == b[1], (name, "bot", i)
    g["seqs"] = [(seqname, 0, length, 0, len(tops), 0, len(bots))]
    if dna is not None:
        g["dna"] = dna
    fix_parse_info(g)
    return g


def random_multiseq_alignment(seed, n_genomes=6, max_children=3, root_len=400, max_seqs=4, root_children=0):
    """Independent random alignment generator for tests (not halRandGen): irregular segment lengths, several
    sequences per genome, inversions, insertions, deletions, duplications (paralogy rings).  Returns the genome
    dict list for write_hgx.  Invariants kept: segments never span sequences, child top segment length == parent
    bottom segment length, bottom child slot points at one (canonical) member of the ring."""
    import random
    rnd = random.Random(seed)
    parents = [-1]
    children = [[]]
    for g in range(1, n_genomes):
        cands = [p for p in range(g) if len(children[p]) < max_children]
        p = 0 if g <= root_children else rnd.choice(cands)  # (root_children: a star at the root, beyond max_children)
        parents.append(p)
        children.append([])
        children[p].append(g)

    def cut(total, lo, hi):
        out, pos = [], 0
        while pos < total:
            l = min(total - pos, rnd.randint(lo, hi))
            out.append(l)
            pos += l
        return out

    genomes = [None] * n_genomes
    bot_lens = {}
    for g in range(n_genomes):
        p = parents[g]
        top = []  # (len, parentIdx, rev)
        if p < 0:
            total = root_len
        else:
            pl = bot_lens[p]
            k = max(3, int(len(pl) * rnd.uniform(0.7, 1.4)))
            i = 0
            for _ in range(k):
                r = rnd.random()
                if r < 0.15:
                    top.append((rnd.randint(1, 25), NULL, False))      # insertion
                elif r < 0.35:
                    j = rnd.randrange(len(pl))
                    top.append((pl[j], j, rnd.random() < 0.4))         # transposition / duplication
                else:
                    top.append((pl[i % len(pl)], i % len(pl), rnd.random() < 0.3))
                    i += rnd.choice([1, 1, 1, 2])                      # occasional deletion
            total = sum(t[0] for t in top)
        tb = [0]
        for t in top:
            tb.append(tb[-1] + t[0])
        nseq = rnd.randint(1, max_seqs)
        if p < 0:
            sb = sorted(set([0, total] + [rnd.randrange(1, total) for _ in range(nseq - 1)]))
        else:
            inner = tb[1:-1]
            sb = sorted(set([0, total] + (rnd.sample(inner, min(len(inner), nseq - 1)) if inner else [])))
        bstarts = []
        if children[g]:
            for a, b in zip(sb[:-1], sb[1:]):
                pos = a
                for l in cut(b - a, 3, 30):
                    bstarts.append(pos)
                    pos += l
        bl = [(bstarts[i + 1] if i + 1 < len(bstarts) else total) - bstarts[i] for i in range(len(bstarts))]
        bot_lens[g] = bl
        name = "G%d" % g
        gd = {"name": name, "parent": p, "children": children[g], "branch": rnd.randint(0, 2)}
        gd["tStart"] = tb if top else [total]
        gd["tParent"] = [t[1] for t in top]
        gd["tParentRev"] = [1 if t[2] else 0 for t in top]
        gd["tParalogy"] = [NULL] * len(top)
        gd["bStart"] = bstarts + [total]
        gd["bChild"] = [[NULL] * len(bstarts) for _ in children[g]]
        gd["bChildRev"] = [[0] * len(bstarts) for _ in children[g]]
        seqs = []
        for si, (a, b) in enumerate(zip(sb[:-1], sb[1:])):
            ti = [i for i in range(len(top)) if a <= tb[i] < b]
            bi = [i for i in range(len(bstarts)) if a <= bstarts[i] < b]
            seqs.append(("%s_chr%d" % (name, si), a, b - a, ti[0] if ti else 0, len(ti), bi[0] if bi else 0, len(bi)))
        gd["seqs"] = seqs
        gd["dna"] = "".join(rnd.choice("ACGTacgtN") for _ in range(total))
        genomes[g] = gd
        if p >= 0:
            P = genomes[p]
            slot = P["children"].index(g)
            ring = {}
            for i, (l, pi, rev) in enumerate(top):
                if pi != NULL:
                    ring.setdefault(pi, []).append(i)
            for pi, members in ring.items():
                canon = rnd.choice(members)
                P["bChild"][slot][pi] = canon
                P["bChildRev"][slot][pi] = gd["tParentRev"][canon]
                if len(members) > 1:
                    for a, b in zip(members, members[1:] + members[:1]):
                        gd["tParalogy"][a] = b
        fix_parse_info(gd)
    return genomes


def scale_alignment(genomes, k):
    """The same alignment with every coordinate multiplied by k (segment and sequence boundaries; links are untouched) and
    without DNA: a few thousand segments then cover genomes of more than 2^31 bases, the case the int64 tables exist for."""
    out = []
    for g in genomes:
        h = dict(g)
        h["tStart"] = [x * k for x in g["tStart"]]
        h["bStart"] = [x * k for x in g["bStart"]]
        h["seqs"] = [(s[0], s[1] * k, s[2] * k, s[3], s[4], s[5], s[6]) for s in g["seqs"]]
        h["dna"] = ""
        out.append(h)
    return out
