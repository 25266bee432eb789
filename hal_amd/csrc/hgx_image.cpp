// HGX flat image I/O, Newick text, structural validation.  See hgx_image.hpp.
#include "hgx_image.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>

namespace hgx {


// api/impl/halCommon.cpp:224-235: a,c,g,t,n -> 0..4, upper case +8, everything else 'n'
const uint8_t dnaPackMap[256] = {
    4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,  4, 4,
    4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 8, 4, 9, 4, 4, 4, 10, 4, 4,
    4, 4, 4, 4, 12, 4, 4, 4, 4, 4, 11, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4,  4, 4,
    4, 4, 4, 4, 4,  3, 4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,  4, 4,
    4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,  4, 4,
    4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,  4, 4,
    4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
const char dnaUnpackMap[16] = {'a', 'c', 'g', 't', 'n', '\0', '\0', '\0', 'A', 'C', 'G', 'T', 'N', '\0', '\0', '\0'};

void packDna(const std::string &s, std::vector<uint8_t> &out) {
    out.assign((s.size() + 1) / 2, 0);
    for (size_t i = 0; i < s.size(); ++i) {
        uint8_t code = dnaPackMap[(uint8_t)s[i]];
        out[i >> 1] |= (i & 1) ? code : (uint8_t)(code << 4);
    }
}

std::string Image::buildNewick() const {
    std::function<void(int, std::string &)> rec = [&](int g, std::string &out) {
        const GenomeTables &G = genomes[(size_t)g];
        if (!G.children.empty()) {
            out += '(';
            for (size_t k = 0; k < G.children.size(); ++k) {
                if (k)
                    out += ',';
                rec(G.children[k], out);
            }
            out += ')';
        }
        out += G.name;
        if (G.parent >= 0) {
            char buf[64];
            snprintf(buf, sizeof buf, ":%g", G.branchLength);
            out += buf;
        }
    };
    std::string s;
    int r = root();
    if (r >= 0)
        rec(r, s);
    s += ';';
    return s;
}

void Image::validate() const {
    auto fail = [](const std::string &m) { throw std::runtime_error("invalid alignment image: " + m); };
    // the tree first: every index below is used as a subscript (a corrupt file must end here, not in a crash, an
    // out-of-bounds read on the device or a ring walk that never comes back)
    const int ng = (int)genomes.size();
    int roots = 0;
    for (int gi = 0; gi < ng; ++gi) {
        const GenomeTables &G = genomes[(size_t)gi];
        if (G.parent < -1 || G.parent >= ng || G.parent == gi)
            fail(G.name + ": parent genome out of range");
        roots += G.parent < 0 ? 1 : 0;
        for (int c : G.children)
            if (c < 0 || c >= ng || c == gi || genomes[(size_t)c].parent != gi)
                fail(G.name + ": child genome out of range or not pointing back");
        if (G.parent >= 0 && genomes[(size_t)G.parent].childSlotOf(gi) < 0)
            fail(G.name + ": not a child of its parent");
        if (G.numTop < 0 || G.numBot < 0 || G.totalLength < 0)
            fail(G.name + ": negative size");
    }
    if (ng > 0 && roots != 1)
        fail("the tree needs exactly one root genome");
    for (int gi = 0; gi < ng; ++gi) { // no cycles: every genome reaches the root
        int steps = 0;
        for (int x = gi; x >= 0; x = genomes[(size_t)x].parent)
            if (++steps > ng)
                fail(genomes[(size_t)gi].name + ": parent links form a cycle");
    }
    // the sizes of every genome's tables first, one after the other: the checks below run genome by genome on several threads
    // and index into the parent's and the children's tables as well
    for (int gi = 0; gi < ng; ++gi) {
        const GenomeTables &G = genomes[(size_t)gi];
        if ((int64_t)G.tStart.size() != G.numTop + 1 || (int64_t)G.bStart.size() != G.numBot + 1)
            fail(G.name + ": start table size");
        if ((int64_t)G.tParent.size() != G.numTop || (int64_t)G.tParalogy.size() != G.numTop || (int64_t)G.tBotParse.size() != G.numTop ||
            (int64_t)G.tParentRev.size() != G.numTop || (int64_t)G.bTopParse.size() != G.numBot)
            fail(G.name + ": segment table size");
        if (G.bChild.size() < G.children.size())
            fail(G.name + ": child link table size");
        for (size_t k = 0; k < G.bChild.size(); ++k)
            if ((int64_t)G.bChild[k].size() != G.numBot || k >= G.bChildRev.size() || (int64_t)G.bChildRev[k].size() != G.numBot)
                fail(G.name + ": child link table size");
        if (!G.dna.empty() && (int64_t)G.dna.size() != (G.totalLength + 1) / 2)
            fail(G.name + ": DNA is not (length + 1) / 2 bytes");
    }
    forEachGenome(genomes.size(), [&](size_t gi) {
        const GenomeTables &G = genomes[gi];
        if (G.parent < 0)
            for (int64_t i = 0; i < G.numTop; ++i)
                if (G.tParent[(size_t)i] != NULL_INDEX)
                    fail(G.name + ": the root genome has a parent link");
        // paralogy rings: in range for every top segment (with or without a parent link) and closed — following the links
        // from any member comes back to it (the ring walks on the device have no other way to stop)
        {
            std::vector<uint8_t> seen((size_t)G.numTop, 0);
            for (int64_t i = 0; i < G.numTop; ++i) {
                const int64_t n = G.tParalogy[(size_t)i];
                if (n == NULL_INDEX)
                    continue;
                if (n < 0 || n >= G.numTop)
                    fail(G.name + ": paralogy index out of range");
                if (seen[(size_t)i])
                    continue;
                int64_t x = i, steps = 0;
                do {
                    seen[(size_t)x] = 1;
                    x = G.tParalogy[(size_t)x];
                    if (x == NULL_INDEX || x < 0 || x >= G.numTop || ++steps > G.numTop)
                        fail(G.name + ": paralogy ring is not closed");
                } while (x != i && !seen[(size_t)x]);
                if (x != i)
                    fail(G.name + ": paralogy ring is not closed");
            }
        }
        if (G.numTop > 0 && (G.tStart[0] != 0 || G.tStart[(size_t)G.numTop] != G.totalLength))
            fail(G.name + ": top tiling does not cover the genome");
        if (G.numBot > 0 && (G.bStart[0] != 0 || G.bStart[(size_t)G.numBot] != G.totalLength))
            fail(G.name + ": bottom tiling does not cover the genome");
        for (int64_t i = 0; i < G.numTop; ++i)
            if (G.tStart[(size_t)i + 1] <= G.tStart[(size_t)i])
                fail(G.name + ": empty or unsorted top segment");
        for (int64_t i = 0; i < G.numBot; ++i)
            if (G.bStart[(size_t)i + 1] <= G.bStart[(size_t)i])
                fail(G.name + ": empty or unsorted bottom segment");
        if (G.bChild.size() != G.children.size() || G.bChildRev.size() != G.children.size())
            fail(G.name + ": child slot count");
        // halValidate.cpp:103-172: parent link in range, same length, paralogs share the parent
        if (G.parent >= 0) {
            const GenomeTables &P = genomes[(size_t)G.parent];
            int slot = P.childSlotOf((int)gi);
            if (slot < 0)
                fail(G.name + ": not a child of its parent");
            for (int64_t i = 0; i < G.numTop; ++i) {
                int64_t p = G.tParent[(size_t)i];
                if (p == NULL_INDEX)
                    continue;
                if (p < 0 || p >= P.numBot)
                    fail(G.name + ": parent index out of range");
                if (P.bStart[(size_t)p + 1] - P.bStart[(size_t)p] != G.tStart[(size_t)i + 1] - G.tStart[(size_t)i])
                    fail(G.name + ": parent segment length differs");
                int64_t n = G.tParalogy[(size_t)i];
                if (n != NULL_INDEX && (n < 0 || n >= G.numTop || G.tParent[(size_t)n] != p))
                    fail(G.name + ": paralogy ring member with a different parent");
            }
        }
        // halValidate.cpp:27-101: child link in range and pointing back
        for (size_t k = 0; k < G.children.size(); ++k) {
            const GenomeTables &C = genomes[(size_t)G.children[k]];
            for (int64_t i = 0; i < G.numBot; ++i) {
                int64_t c = G.bChild[k][(size_t)i];
                if (c == NULL_INDEX)
                    continue;
                if (c < 0 || c >= C.numTop || C.tParent[(size_t)c] != i)
                    fail(G.name + ": child link does not point back");
            }
        }
        // parse indices contain the segment start (halValidate.cpp:55-75,140-160)
        if (G.numTop > 0 && G.numBot > 0) {
            for (int64_t i = 0; i < G.numTop; ++i) {
                int64_t b = G.tBotParse[(size_t)i];
                if (b < 0 || b >= G.numBot || G.bStart[(size_t)b] > G.tStart[(size_t)i] ||
                    G.bStart[(size_t)b + 1] <= G.tStart[(size_t)i])
                    fail(G.name + ": bottom parse index");
            }
            for (int64_t i = 0; i < G.numBot; ++i) {
                int64_t t = G.bTopParse[(size_t)i];
                if (t < 0 || t >= G.numTop || G.tStart[(size_t)t] > G.bStart[(size_t)i] ||
                    G.tStart[(size_t)t + 1] <= G.bStart[(size_t)i])
                    fail(G.name + ": top parse index");
            }
        }
        // halValidate.cpp:174-221: sequences tile the genome and own whole segments
        int64_t pos = 0, topAt = 0, botAt = 0;
        for (const SeqInfo &S : G.seqs) {
            if (S.start != pos || S.length < 0)
                fail(G.name + ": sequences not laid end to end");
            if (S.numTop < 0 || S.numBot < 0 || S.topStart < 0 || S.botStart < 0 || S.topStart + S.numTop > G.numTop ||
                S.botStart + S.numBot > G.numBot)
                fail(G.name + ": sequence segment range out of bounds");
            if (S.numTop > 0 && (S.topStart != topAt || G.tStart[(size_t)S.topStart] != S.start ||
                                 G.tStart[(size_t)(S.topStart + S.numTop)] != S.start + S.length))
                fail(G.name + ": top segments of sequence " + S.name + " do not tile it");
            if (S.numBot > 0 && (S.botStart != botAt || G.bStart[(size_t)S.botStart] != S.start ||
                                 G.bStart[(size_t)(S.botStart + S.numBot)] != S.start + S.length))
                fail(G.name + ": bottom segments of sequence " + S.name + " do not tile it");
            topAt += S.numTop;
            botAt += S.numBot;
            pos += S.length;
        }
        if (pos != G.totalLength)
            fail(G.name + ": sequence lengths do not sum to the genome length");
        if (!G.seqs.empty() && ((G.numTop > 0 && topAt != G.numTop) || (G.numBot > 0 && botAt != G.numBot)))
            fail(G.name + ": segments outside every sequence");
    });
}

// ---- HGX file ----
namespace {
struct Writer {
    FILE *f;
    explicit Writer(const std::string &p) : f(fopen(p.c_str(), "wb")) {
        if (!f)
            throw std::runtime_error("cannot create " + p);
    }
    ~Writer() {
        if (f)
            fclose(f);
    }
    void raw(const void *p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n)
            throw std::runtime_error("short write");
    }
    void s64(int64_t v) {
        raw(&v, 8);
    }
    void pad(size_t n) {
        static const char z[8] = {0};
        raw(z, (8 - n % 8) % 8);
    }
    void str(const std::string &s) {
        s64((int64_t)s.size());
        raw(s.data(), s.size());
        pad(s.size());
    }
    void a64(const std::vector<int64_t> &v) {
        raw(v.data(), v.size() * 8);
    }
    void a8(const std::vector<uint8_t> &v) {
        raw(v.data(), v.size());
        pad(v.size());
    }
};
struct FReader {
    FILE *f;
    explicit FReader(const std::string &p) : f(fopen(p.c_str(), "rb")) {
        if (!f)
            throw std::runtime_error("cannot open " + p);
    }
    ~FReader() {
        if (f)
            fclose(f);
    }
    void raw(void *p, size_t n) {
        if (n && fread(p, 1, n, f) != n)
            throw std::runtime_error("truncated HGX image");
    }
    int64_t s64() {
        int64_t v;
        raw(&v, 8);
        return v;
    }
    void pad(size_t n) {
        char z[8];
        raw(z, (8 - n % 8) % 8);
    }
    std::string str() {
        int64_t n = s64();
        if (n < 0 || n > (1 << 28))
            throw std::runtime_error("corrupt HGX string length");
        std::string s((size_t)n, '\0');
        raw(&s[0], (size_t)n);
        pad((size_t)n);
        return s;
    }
    void a64(std::vector<int64_t> &v, size_t n) {
        v.resize(n);
        raw(v.data(), n * 8);
    }
    void a8(std::vector<uint8_t> &v, size_t n) {
        v.resize(n);
        raw(v.data(), n);
        pad(n);
    }
};
} // namespace

void writeImage(const Image &img, const std::string &path) {
    Writer w(path);
    w.raw("HGXIMG01", 8);
    w.s64((int64_t)img.genomes.size());
    w.str(img.newick);
    for (const GenomeTables &G : img.genomes) {
        w.str(G.name);
        w.s64(G.parent);
        w.s64((int64_t)G.children.size());
        for (int c : G.children)
            w.s64(c);
        w.s64(G.totalLength);
        w.s64((int64_t)G.seqs.size());
        w.s64(G.numTop);
        w.s64(G.numBot);
        for (const SeqInfo &S : G.seqs) {
            w.str(S.name);
            w.s64(S.start);
            w.s64(S.length);
            w.s64(S.topStart);
            w.s64(S.numTop);
            w.s64(S.botStart);
            w.s64(S.numBot);
        }
        w.a64(G.tStart);
        w.a64(G.tParent);
        w.a64(G.tParalogy);
        w.a64(G.tBotParse);
        w.a8(G.tParentRev);
        w.a64(G.bStart);
        w.a64(G.bTopParse);
        for (size_t k = 0; k < G.children.size(); ++k) {
            w.a64(G.bChild[k]);
            w.a8(G.bChildRev[k]);
        }
        w.s64((int64_t)G.dna.size());
        w.a8(G.dna);
    }
}

Image readImage(const std::string &path) {
    FReader r(path);
    char magic[8];
    r.raw(magic, 8);
    if (memcmp(magic, "HGXIMG01", 8) != 0)
        throw std::runtime_error(path + ": not an HGX image");
    Image img;
    // counts are checked against the file's size before anything is resized by them
    int64_t fileSize = 0;
    {
        const long at = ftell(r.f);
        fseek(r.f, 0, SEEK_END);
        fileSize = (int64_t)ftell(r.f);
        fseek(r.f, at, SEEK_SET);
    }
    auto sane = [&](int64_t n, int64_t bytesEach, const char *what) {
        if (n < 0 || n > fileSize / std::max<int64_t>(1, bytesEach))
            throw std::runtime_error(path + ": corrupt HGX image (" + what + ")");
    };
    int64_t ng = r.s64();
    sane(ng, 64, "genome count");
    img.newick = r.str();
    img.genomes.resize((size_t)ng);
    for (GenomeTables &G : img.genomes) {
        G.name = r.str();
        const int64_t parent = r.s64();
        if (parent < -1 || parent >= ng)
            throw std::runtime_error(path + ": corrupt HGX image (parent genome)");
        G.parent = (int)parent;
        int64_t nc = r.s64();
        sane(nc, 8, "child count");
        G.children.resize((size_t)nc);
        for (int &c : G.children) {
            const int64_t v = r.s64();
            if (v < 0 || v >= ng)
                throw std::runtime_error(path + ": corrupt HGX image (child genome)");
            c = (int)v;
        }
        G.totalLength = r.s64();
        int64_t ns = r.s64();
        G.numTop = r.s64();
        G.numBot = r.s64();
        sane(ns, 56, "sequence count");
        sane(G.numTop, 33, "top segment count");
        sane(G.numBot, 16, "bottom segment count");
        if (G.totalLength < 0)
            throw std::runtime_error(path + ": corrupt HGX image (genome length)");
        G.seqs.resize((size_t)ns);
        for (SeqInfo &S : G.seqs) {
            S.name = r.str();
            S.start = r.s64();
            S.length = r.s64();
            S.topStart = r.s64();
            S.numTop = r.s64();
            S.botStart = r.s64();
            S.numBot = r.s64();
        }
        r.a64(G.tStart, (size_t)G.numTop + 1);
        r.a64(G.tParent, (size_t)G.numTop);
        r.a64(G.tParalogy, (size_t)G.numTop);
        r.a64(G.tBotParse, (size_t)G.numTop);
        r.a8(G.tParentRev, (size_t)G.numTop);
        r.a64(G.bStart, (size_t)G.numBot + 1);
        r.a64(G.bTopParse, (size_t)G.numBot);
        G.bChild.resize((size_t)nc);
        G.bChildRev.resize((size_t)nc);
        for (int64_t k = 0; k < nc; ++k) {
            r.a64(G.bChild[(size_t)k], (size_t)G.numBot);
            r.a8(G.bChildRev[(size_t)k], (size_t)G.numBot);
        }
        int64_t nd = r.s64();
        sane(nd, 1, "DNA size");
        r.a8(G.dna, (size_t)nd);
    }
    return img;
}

Image openAlignmentFile(const std::string &path) {
    char head[8] = {0};
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f)
            throw std::runtime_error("cannot open " + path);
        size_t n = fread(head, 1, 8, f);
        fclose(f);
        if (n != 8)
            throw std::runtime_error(path + ": too short to be an alignment file");
    }
    if (memcmp(head, "HGXIMG01", 8) == 0)
        return readImage(path);
    if (memcmp(head, "HAL-MMAP", 8) == 0)
        return readMmapHal(path);
    if (memcmp(head, "\x89HDF\r\n\x1a\n", 8) == 0)
        return readHdf5Hal(path);
    throw std::runtime_error(path + ": unknown alignment file format");
}

} // namespace hgx
