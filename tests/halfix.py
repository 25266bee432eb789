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
