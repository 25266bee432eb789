// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the HAL column/block traversal hot path, written from the
// reference's algorithm (every function cites the reference file:line it
// follows).  It deliberately keeps the reference's data structures
// (std::list / std::set of heap-allocated mapped segments, per-step iterator
// objects, in-place clipping of set members) so that it can serve as the
// checker for the HIP product path and as the `cpu_baseline` leg of bench.py.
//
// NOTHING in hal_amd/ may include, link or execute this code; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
//
// Parity pinning: see oracle/README.md (golden vectors of the reference's own
// tests: liftover/tests/expected/*.bed, liftover/tests/halLiftoverTests.cpp
// literal strings, maf/tests/expected/*.maf).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

typedef int64_t i64;
typedef uint64_t u64;
typedef uint8_t u8;

static const i64 NULL_INDEX = -1; // api/impl/halCommon.cpp:18

// One sequence of a genome (api/mmap_impl/mmapSequenceData.h:20-30).
struct Sequence {
    std::string name;
    i64 start;    // first base in genome coordinates
    i64 length;
    i64 topStart; // index of first top segment
    i64 numTop;
    i64 botStart;
    i64 numBot;
};

// One genome: its top tiling (links to the parent) and bottom tiling (links to
// each child).  Arrays follow api/mmap_impl/mmapTopSegmentData.h:40-44 and
// api/mmap_impl/mmapBottomSegmentData.h:35-52, transposed.
struct Genome {
    std::string name;
    int parent;                // genome index or -1
    std::vector<int> children; // child slot k -> genome index (Newick order)
    i64 totalLength;
    std::vector<Sequence> seqs;
    i64 numTop;
    std::vector<i64> tStart; // numTop+1 (sentinel = totalLength)
    std::vector<i64> tParent;
    std::vector<i64> tParalogy;
    std::vector<i64> tBotParse;
    std::vector<u8> tParentRev;
    i64 numBot;
    std::vector<i64> bStart; // numBot+1
    std::vector<i64> bTopParse;
    std::vector<std::vector<i64>> bChild;   // [slot][seg]
    std::vector<std::vector<u8>> bChildRev; // [slot][seg]
    std::vector<u8> dna;                    // 2 bases / byte, even index in the high nibble
    i64 childSlotOf(int childGenome) const {
        for (size_t k = 0; k < children.size(); ++k)
            if (children[k] == childGenome)
                return (i64)k;
        return NULL_INDEX;
    }
    const Sequence *seqBySite(i64 pos) const {
        // api/mmap_impl/mmapGenomeSiteMap.cpp:99-113 (BST) == binary search on start[]
        size_t lo = 0, hi = seqs.size();
        while (hi - lo > 1) {
            size_t mid = (lo + hi) / 2;
            if (seqs[mid].start <= pos)
                lo = mid;
            else
                hi = mid;
        }
        if (seqs.empty() || pos < seqs[lo].start || pos >= seqs[lo].start + seqs[lo].length)
            return nullptr;
        return &seqs[lo];
    }
    const Sequence *seqByName(const std::string &n) const {
        for (auto &s : seqs)
            if (s.name == n)
                return &s;
        return nullptr;
    }
};

struct Alignment {
    std::string newick;
    std::vector<Genome> genomes;
    int genomeByName(const std::string &n) const {
        for (size_t i = 0; i < genomes.size(); ++i)
            if (genomes[i].name == n)
                return (int)i;
        return -1;
    }
    int root() const {
        for (size_t i = 0; i < genomes.size(); ++i)
            if (genomes[i].parent < 0)
                return (int)i;
        return -1;
    }
};

// ---- HGX flat image reader (independent of hal_amd's writer; format in DESIGN.md) ----
struct Reader {
    FILE *f;
    explicit Reader(const std::string &path) : f(fopen(path.c_str(), "rb")) {
        if (!f)
            throw std::runtime_error("cannot open " + path);
    }
    ~Reader() {
        if (f)
            fclose(f);
    }
    void raw(void *p, size_t n) {
        if (n && fread(p, 1, n, f) != n)
            throw std::runtime_error("short read in HGX image");
    }
    i64 s64() {
        i64 v;
        raw(&v, 8);
        return v;
    }
    void pad(size_t n) {
        size_t r = (8 - n % 8) % 8;
        char buf[8];
        raw(buf, r);
    }
    std::string str() {
        i64 n = s64();
        std::string s((size_t)n, '\0');
        raw(&s[0], (size_t)n);
        pad((size_t)n);
        return s;
    }
    void arr64(std::vector<i64> &v, size_t n) {
        v.resize(n);
        raw(v.data(), n * 8);
    }
    void arr8(std::vector<u8> &v, size_t n) {
        v.resize(n);
        raw(v.data(), n);
        pad(n);
    }
};

inline Alignment loadImage(const std::string &path) {
    Reader r(path);
    char magic[8];
    r.raw(magic, 8);
    if (memcmp(magic, "HGXIMG01", 8) != 0)
        throw std::runtime_error("not an HGX image: " + path);
    Alignment a;
    i64 ng = r.s64();
    a.newick = r.str();
    a.genomes.resize((size_t)ng);
    for (i64 g = 0; g < ng; ++g) {
        Genome &G = a.genomes[(size_t)g];
        G.name = r.str();
        G.parent = (int)r.s64();
        i64 nc = r.s64();
        G.children.resize((size_t)nc);
        for (i64 c = 0; c < nc; ++c)
            G.children[(size_t)c] = (int)r.s64();
        G.totalLength = r.s64();
        i64 ns = r.s64();
        G.numTop = r.s64();
        G.numBot = r.s64();
        G.seqs.resize((size_t)ns);
        for (i64 s = 0; s < ns; ++s) {
            Sequence &S = G.seqs[(size_t)s];
            S.name = r.str();
            S.start = r.s64();
            S.length = r.s64();
            S.topStart = r.s64();
            S.numTop = r.s64();
            S.botStart = r.s64();
            S.numBot = r.s64();
        }
        r.arr64(G.tStart, (size_t)G.numTop + 1);
        r.arr64(G.tParent, (size_t)G.numTop);
        r.arr64(G.tParalogy, (size_t)G.numTop);
        r.arr64(G.tBotParse, (size_t)G.numTop);
        r.arr8(G.tParentRev, (size_t)G.numTop);
        r.arr64(G.bStart, (size_t)G.numBot + 1);
        r.arr64(G.bTopParse, (size_t)G.numBot);
        G.bChild.resize((size_t)nc);
        G.bChildRev.resize((size_t)nc);
        for (i64 c = 0; c < nc; ++c) {
            r.arr64(G.bChild[(size_t)c], (size_t)G.numBot);
            r.arr8(G.bChildRev[(size_t)c], (size_t)G.numBot);
        }
        i64 nd = r.s64();
        r.arr8(G.dna, (size_t)nd);
    }
    return a;
}

} // namespace orc
