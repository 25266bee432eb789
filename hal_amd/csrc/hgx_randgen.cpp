// Synthetic alignment generator with the semantics of the reference's halRandGen
// (randgen/halRandGen.cpp:34-37 presets, :110 rng; api/tests/halRandomData.cpp:62-348;
// api/tests/halRandNumberGen.h).  It builds the flat image directly (no storage back end) and is the
// workload generator of every BASELINE config; for equal seed/options it produces the same tree,
// tilings, parent/child/paralogy links and DNA as halRandGen (pinned through the reference's golden
// liftover/MAF outputs in tests/).
#include "hgx_image.hpp"
#include <cstring>
#include <cmath>
#include <deque>
#include <random>

namespace hgx {

namespace {

// api/tests/halRandNumberGen.h:23-115 (non-test mode: std::mt19937 + a fresh
// uniform_real_distribution<double> per draw)
struct Rng {
    std::mt19937 rng;
    explicit Rng(int seed) {
        rng.seed(seed); // halRandNumberGen.h:53-55 (_seed member is 0 there, so always seeded)
    }
    double getRand() {
        std::uniform_real_distribution<double> dist;
        return dist(rng);
    }
    // :74-81 — declared to return int in the reference, so the value is truncated
    int getRandDouble(double minVal, double maxVal) {
        if (maxVal < minVal)
            maxVal = minVal;
        return (int)((getRand() * (maxVal - minVal)) + minVal);
    }
    // :86-98
    int getRandInt(int minVal, int maxVal) {
        if (maxVal < minVal)
            maxVal = minVal;
        double rnum = getRand() * double(maxVal - minVal);
        if ((rnum - floor(rnum)) >= 0.5)
            return minVal + int(ceil(rnum));
        return minVal + int(floor(rnum));
    }
};

inline bool exponEvent(Rng &rng, double mu) { // halRandomData.cpp:19-21
    return rng.getRand() <= (1.0 - exp(-mu));
}
inline char randDNA(Rng &rng) { // halRandomData.cpp:23-35
    switch (rng.getRandInt(0, 3)) {
    case 0:
        return 'A';
    case 1:
        return 'C';
    case 2:
        return 'G';
    default:
        return 'T';
    }
}
// withDna == 2: bases from a separate cheap generator (splitmix64), so that the main stream — and with it the tree, the
// tilings and every link — is the one of withDna == 0, and a 1 Gb alignment gets its DNA in seconds.  Same model as
// halRandomData.cpp (uniform bases, per-base substitution with probability 1 - exp(-branchLength), reverse complement on
// inverted segments), not the same draws.
struct FastDna {
    uint64_t s;
    explicit FastDna(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void fill(char *dst, int64_t n) {
        static const char B[4] = {'A', 'C', 'G', 'T'};
        int64_t i = 0;
        while (i < n) {
            uint64_t r = next();
            for (int k = 0; k < 32 && i < n; ++k, r >>= 2)
                dst[i++] = B[r & 3];
        }
    }
    // substitutes each base with probability p (threshold on 32-bit draws)
    void mutate(char *dst, int64_t n, double p) {
        static const char B[4] = {'A', 'C', 'G', 'T'};
        if (p <= 0)
            return;
        const uint64_t thr = p >= 1 ? (1ull << 32) : (uint64_t)(p * 4294967296.0);
        for (int64_t i = 0; i < n; i += 2) {
            const uint64_t r = next();
            if ((r & 0xFFFFFFFFull) < thr)
                dst[i] = B[(r >> 60) & 3];
            if (i + 1 < n && ((r >> 28) & 0xFFFFFFFFull) < thr)
                dst[i + 1] = B[(r >> 62) & 3];
        }
    }
};
inline char complement(char c) { // api/inc/halCommon.h:45-75
    switch (c) {
    case 'A':
        return 'T';
    case 'a':
        return 't';
    case 'C':
        return 'G';
    case 'c':
        return 'g';
    case 'G':
        return 'C';
    case 'g':
        return 'c';
    case 'T':
        return 'A';
    case 't':
        return 'a';
    default:
        return c;
    }
}

struct Dim {
    uint64_t botSegSize = 0, topSegSize = 0, length = 0;
};

} // namespace

bool randPreset(const std::string &name, RandOptions &o) { // halRandGen.cpp:34-37
    int seed = o.seed;
    const int dna = o.withDna;
    if (name == "small")
        o = RandOptions{0.75, 0.1, 2, 5, 250, 1000, 5, 10};
    else if (name == "medium")
        o = RandOptions{1.25, 0.7, 8, 20, 500, 2000, 100, 500};
    else if (name == "big")
        o = RandOptions{2.00, 0.7, 20, 50, 1000, 8000, 400, 5000};
    else if (name == "large")
        o = RandOptions{2.00, 1.0, 50, 100, 5000, 10000, 10000, 50000};
    else
        return false;
    o.seed = seed;
    o.withDna = dna;
    return true;
}

Image createRandomAlignment(const RandOptions &opt) {
    if (opt.meanDegree <= 0.0 || opt.maxBranchLength <= 0.0 || opt.minGenomes == 0 || opt.minGenomes > opt.maxGenomes ||
        opt.minSegmentLength == 0 || opt.minSegmentLength > opt.maxSegmentLength || opt.minSegments == 0 ||
        opt.minSegments > opt.maxSegments)
        throw std::runtime_error("createRandomAlignment: invalid options"); // halRandomData.cpp:65-88
    Rng rng(opt.seed);
    Image img;

    // ---- tree (halRandomData.cpp:95-134) ----
    img.genomes.emplace_back();
    img.genomes[0].name = "Genome_0";
    {
        std::deque<int> queue;
        queue.push_front(0);
        uint64_t genomeCount = 1;
        while (!queue.empty()) {
            int g = queue.back();
            queue.pop_back();
            uint64_t numChildren = (uint64_t)(rng.getRandDouble(0.0, 2.0 * opt.meanDegree) + 0.5);
            if (genomeCount + numChildren >= opt.maxGenomes)
                numChildren = opt.maxGenomes - genomeCount;
            if (genomeCount + numChildren < opt.minGenomes)
                numChildren = opt.minGenomes;
            if (numChildren > 100000)
                throw std::runtime_error("createRandomAlignment: runaway tree (the reference generator never terminates for "
                                         "these options/seed; see SURVEY 8(d))");
            for (uint64_t i = 0; i < numChildren; ++i) {
                GenomeTables child;
                child.name = "Genome_" + std::to_string(genomeCount++);
                child.parent = g;
                child.branchLength = rng.getRandDouble(1e-5, opt.maxBranchLength);
                int id = (int)img.genomes.size();
                img.genomes.push_back(child);
                img.genomes[(size_t)g].children.push_back(id);
                queue.push_front(id);
            }
        }
    }
    img.newick = img.buildNewick();

    // ---- dimensions (halRandomData.cpp:136-227), BFS order ----
    std::vector<Dim> dims(img.genomes.size());
    {
        std::deque<int> queue;
        queue.push_front(0);
        while (!queue.empty()) {
            int g = queue.back();
            queue.pop_back();
            GenomeTables &G = img.genomes[(size_t)g];
            Dim &D = dims[(size_t)g];
            D.botSegSize = (uint64_t)rng.getRandInt((int)opt.minSegmentLength, (int)opt.maxSegmentLength);
            uint64_t numBottom = (uint64_t)rng.getRandInt((int)opt.minSegments, (int)opt.maxSegments);
            uint64_t length = numBottom * D.botSegSize;
            uint64_t numTop = 0;
            if (G.parent >= 0) {
                D.topSegSize = dims[(size_t)G.parent].botSegSize;
                numTop = length / D.topSegSize + (length % D.topSegSize != 0 ? 1 : 0);
            }
            if (G.children.empty())
                numBottom = 0;
            if (numBottom == 0 && numTop == 0)
                length = 0;
            D.length = length;
            G.totalLength = (int64_t)length;
            G.numTop = (int64_t)numTop;
            G.numBot = (int64_t)numBottom;
            SeqInfo S;
            S.name = G.name + "_seq";
            S.start = 0;
            S.length = (int64_t)length;
            S.topStart = 0;
            S.numTop = G.numTop;
            S.botStart = 0;
            S.numBot = G.numBot;
            G.seqs.push_back(S);
            size_t nc = G.children.size();
            G.bStart.resize(numBottom + 1);
            G.bTopParse.assign(numBottom, NULL_INDEX);
            G.bChild.assign(nc, std::vector<int64_t>(numBottom, NULL_INDEX));
            G.bChildRev.assign(nc, std::vector<uint8_t>(numBottom, 0));
            for (uint64_t i = 0; i < numBottom; ++i) {
                G.bStart[i] = (int64_t)(i * D.botSegSize);
                if (numTop > 0)
                    G.bTopParse[i] = (int64_t)((i * D.botSegSize) / D.topSegSize);
            }
            G.bStart[numBottom] = (int64_t)length;
            G.tStart.resize(numTop + 1);
            G.tParent.assign(numTop, NULL_INDEX);
            G.tParalogy.assign(numTop, NULL_INDEX);
            G.tBotParse.assign(numTop, NULL_INDEX);
            G.tParentRev.assign(numTop, 0);
            for (uint64_t i = 0; i < numTop; ++i) {
                G.tStart[i] = (int64_t)(i * D.topSegSize);
                if (numBottom > 0)
                    G.tBotParse[i] = (int64_t)((i * D.topSegSize) / D.botSegSize);
            }
            G.tStart[numTop] = (int64_t)length;
            for (int c : G.children)
                queue.push_front(c);
        }
    }

    // ---- segments + DNA (halRandomData.cpp:37-60,229-348), BFS order ----
    std::vector<std::string> dna(img.genomes.size());
    {
        std::deque<int> queue;
        queue.push_front(0);
        std::string buffer;
        while (!queue.empty()) {
            int g = queue.back();
            queue.pop_back();
            GenomeTables &G = img.genomes[(size_t)g];
            std::string &seq = dna[(size_t)g];
            const bool slowDna = opt.withDna == 1, fastDna = opt.withDna == 2;
            FastDna fd((uint64_t)opt.seed * 1000003ull + (uint64_t)g);
            if (opt.withDna)
                seq.resize((size_t)G.totalLength);
            if (G.parent < 0) {
                if (slowDna)
                    for (int64_t i = 0; i < G.totalLength; ++i)
                        seq[(size_t)i] = randDNA(rng);
                else if (fastDna)
                    fd.fill(&seq[0], G.totalLength);
            } else {
                GenomeTables &P = img.genomes[(size_t)G.parent];
                const std::string &pseq = dna[(size_t)G.parent];
                int slot = P.childSlotOf(g);
                double branchLength = G.branchLength;
                int64_t numTopSegs = G.numTop, numBotSegs = P.numBot;
                // edgeSet of halRandomData.cpp:324-342, kept as first/last child per parent segment
                std::vector<int64_t> firstChild((size_t)numBotSegs, NULL_INDEX), lastChild((size_t)numBotSegs, NULL_INDEX);
                for (int64_t i = 0; i < numTopSegs; ++i) {
                    int64_t parentIdx = i;
                    if (parentIdx >= numBotSegs || exponEvent(rng, branchLength))
                        parentIdx = rng.getRandInt(0, (int)(numBotSegs - 1));
                    else if (exponEvent(rng, branchLength) && exponEvent(rng, branchLength))
                        parentIdx = NULL_INDEX;
                    if (parentIdx == numBotSegs - 1 || i == numTopSegs - 1)
                        parentIdx = NULL_INDEX;
                    G.tParent[(size_t)i] = parentIdx;
                    int64_t tstart = G.tStart[(size_t)i], tlen = G.tStart[(size_t)i + 1] - tstart;
                    if (parentIdx == NULL_INDEX) {
                        if (slowDna)
                            for (int64_t j = 0; j < tlen; ++j)
                                seq[(size_t)(tstart + j)] = randDNA(rng);
                        else if (fastDna)
                            fd.fill(&seq[(size_t)tstart], tlen);
                    } else {
                        bool reversed = exponEvent(rng, branchLength);
                        G.tParentRev[(size_t)i] = reversed;
                        if (fastDna) {
                            int64_t pstart = P.bStart[(size_t)parentIdx];
                            char *d = &seq[(size_t)tstart];
                            if (reversed)
                                for (int64_t j = 0; j < tlen; ++j)
                                    d[j] = complement(pseq[(size_t)(pstart + tlen - 1 - j)]);
                            else
                                memcpy(d, pseq.data() + pstart, (size_t)tlen);
                            fd.mutate(d, tlen, 1.0 - exp(-branchLength));
                        }
                        if (slowDna) {
                            int64_t pstart = P.bStart[(size_t)parentIdx];
                            buffer.assign(pseq, (size_t)pstart, (size_t)tlen);
                            if (reversed) {
                                for (int64_t j = 0; j < tlen; ++j)
                                    buffer[(size_t)j] = complement(pseq[(size_t)(pstart + tlen - 1 - j)]);
                            }
                            for (int64_t j = 0; j < tlen; ++j) // mutateString :37-43
                                if (exponEvent(rng, branchLength))
                                    buffer[(size_t)j] = randDNA(rng);
                            seq.replace((size_t)tstart, (size_t)tlen, buffer);
                        }
                        P.bChild[(size_t)slot][(size_t)parentIdx] = i;
                        P.bChildRev[(size_t)slot][(size_t)parentIdx] = reversed;
                        if (lastChild[(size_t)parentIdx] != NULL_INDEX) {
                            G.tParalogy[(size_t)lastChild[(size_t)parentIdx]] = i;
                            G.tParalogy[(size_t)i] = firstChild[(size_t)parentIdx];
                        } else {
                            firstChild[(size_t)parentIdx] = i;
                        }
                        lastChild[(size_t)parentIdx] = i;
                    }
                }
            }
            for (int c : G.children)
                queue.push_front(c);
        }
    }
    for (size_t g = 0; g < img.genomes.size(); ++g) {
        if (opt.withDna)
            packDna(dna[g], img.genomes[g].dna);
        std::string().swap(dna[g]);
    }
    return img;
}

} // namespace hgx
