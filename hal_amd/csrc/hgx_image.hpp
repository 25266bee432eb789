// Host-side alignment image: the tree plus, per genome, the top/bottom segment tables of a HAL
// alignment flattened to Structure-of-Arrays.  This is the form the mmap HAL reader
// (hgx_mmap_reader.cpp), the synthetic generator (hgx_randgen.cpp) and hgx_create_from_arrays()
// all produce, and the form hgx_device.hip narrows and uploads to HBM.
//
// Reference layouts being flattened (all under /root/reference/api/mmap_impl):
//   mmapTopSegmentData.h:40-44     {startPosition, bottomParseIndex, paralogyIndex, parentIndex, reversed}
//   mmapBottomSegmentData.h:35-52  {startPosition, topParseIndex, childIndex[nc], childReversed[nc]}
//   mmapSequenceData.h:20-30       {startPosition, index, length, top/bottom start index + counts, name}
//   mmapGenome.h:19-46             per-genome counts and array offsets
// A segment's length is next.start - start, so each start table carries one sentinel entry
// (mmapTopSegment.h:78-80, mmapGenome.cpp:141).
#pragma once
#include <algorithm>
#include <atomic>
#include <exception>
#include <thread>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "hgx_host_threads.hpp"

namespace hgx {

// fn(g) for every genome g < n on a few threads (a genome's tables are converted and checked on their own; what fn throws for the
// genome with the smallest index comes out, as if the genomes had been gone through in order)
template <class Fn> void forEachGenome(size_t n, Fn fn) {
    unsigned nt = hostThreads();
    nt = (unsigned)std::min<size_t>(std::max(1u, std::min(nt ? nt : 1u, 16u)), n);
    if (nt <= 1) {
        for (size_t g = 0; g < n; ++g)
            fn(g);
        return;
    }
    std::vector<std::exception_ptr> failed(n);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t g; (g = next.fetch_add(1)) < n;) {
            try {
                fn(g);
            } catch (...) {
                failed[g] = std::current_exception();
            }
        }
    };
    std::vector<std::thread> threads;
    for (unsigned t = 1; t < nt; ++t)
        threads.emplace_back(work);
    work();
    for (std::thread &t : threads)
        t.join();
    for (size_t g = 0; g < n; ++g)
        if (failed[g])
            std::rethrow_exception(failed[g]);
}

static const int64_t NULL_INDEX = -1; // api/impl/halCommon.cpp:18

struct SeqInfo {
    std::string name;
    int64_t start = 0;  // genome coordinate of first base
    int64_t length = 0;
    int64_t topStart = 0, numTop = 0;
    int64_t botStart = 0, numBot = 0;
};

struct GenomeTables {
    std::string name;
    int parent = -1;
    std::vector<int> children; // child slot -> genome id, in Newick order (mmapAlignment.h:145-153)
    double branchLength = 0;   // branch to parent (0 for root)
    int64_t totalLength = 0;
    std::vector<SeqInfo> seqs; // sorted by start
    int64_t numTop = 0, numBot = 0;
    std::vector<int64_t> tStart, tParent, tParalogy, tBotParse;
    std::vector<uint8_t> tParentRev;
    std::vector<int64_t> bStart, bTopParse;
    std::vector<std::vector<int64_t>> bChild;
    std::vector<std::vector<uint8_t>> bChildRev;
    std::vector<uint8_t> dna; // nibble-packed, even base index in the high nibble (halCommon.h:187-196)

    int childSlotOf(int g) const {
        for (size_t k = 0; k < children.size(); ++k)
            if (children[k] == g)
                return (int)k;
        return -1;
    }
    int seqIndexByName(const std::string &n) const {
        for (size_t i = 0; i < seqs.size(); ++i)
            if (seqs[i].name == n)
                return (int)i;
        return -1;
    }
    // sequence containing genome position pos (binary search on start[]; the reference uses a BST,
    // mmapGenomeSiteMap.cpp:99-113, same answer)
    int seqIndexBySite(int64_t pos) const {
        if (seqs.empty())
            return -1;
        size_t lo = 0, hi = seqs.size();
        while (hi - lo > 1) {
            size_t mid = (lo + hi) / 2;
            if (seqs[mid].start <= pos)
                lo = mid;
            else
                hi = mid;
        }
        if (pos < seqs[lo].start || pos >= seqs[lo].start + seqs[lo].length)
            return -1;
        return (int)lo;
    }
};

struct Image {
    std::string newick;
    std::vector<GenomeTables> genomes;

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
    int depthOf(int g) const {
        int d = 0;
        while (genomes[(size_t)g].parent >= 0) {
            g = genomes[(size_t)g].parent;
            ++d;
        }
        return d;
    }
    // lowest common ancestor (api/impl/halCommon.cpp:123-152 computes the same node)
    int lca(int a, int b) const {
        int da = depthOf(a), db = depthOf(b);
        while (da > db) {
            a = genomes[(size_t)a].parent;
            --da;
        }
        while (db > da) {
            b = genomes[(size_t)b].parent;
            --db;
        }
        while (a != b) {
            a = genomes[(size_t)a].parent;
            b = genomes[(size_t)b].parent;
        }
        return a;
    }
    std::string buildNewick() const; // sonLib-style "(kids)label:len;" text
    // structural checks mirroring api/impl/halValidate.cpp:27-251; throws std::runtime_error
    void validate() const;
};

// HGX flat image file ("HGXIMG01"; layout documented in DESIGN.md §3)
void writeImage(const Image &img, const std::string &path);
Image readImage(const std::string &path);

// mmap-format HAL reader (api/mmap_impl/*, versions 1.0 and 1.1)
// Newick subset written by the reference (sonLib stTree_getNewickTreeString): "(a,b)label:len;".
struct NewickNode {
    std::string label;
    double len = 0;
    std::vector<int> kids;
};
struct NewickParser {
    const std::string &s;
    size_t i = 0;
    std::vector<NewickNode> nodes;
    explicit NewickParser(const std::string &str) : s(str) {
    }
    int parse() {
        int id = (int)nodes.size();
        nodes.emplace_back();
        if (i < s.size() && s[i] == '(') {
            ++i;
            for (;;) {
                int k = parse();
                nodes[(size_t)id].kids.push_back(k);
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == ')') {
                    ++i;
                    break;
                }
                throw std::runtime_error("malformed Newick tree in the alignment file");
            }
        }
        size_t b = i;
        while (i < s.size() && s[i] != ':' && s[i] != ',' && s[i] != ')' && s[i] != ';')
            ++i;
        nodes[(size_t)id].label = s.substr(b, i - b);
        if (i < s.size() && s[i] == ':') {
            ++i;
            size_t e = i;
            while (e < s.size() && s[e] != ',' && s[e] != ')' && s[e] != ';')
                ++e;
            nodes[(size_t)id].len = atof(s.substr(i, e - i).c_str());
            i = e;
        }
        return id;
    }
};


Image readMmapHal(const std::string &path);
// HDF5-format HAL (api/hdf5_impl); needs libhdf5 at run time (hgx_hdf5_reader.cpp)
Image readHdf5Hal(const std::string &path);
// auto-detect by magic: "HGXIMG01" or "HAL-MMAP"
Image openAlignmentFile(const std::string &path);

// DNA nibble codec (api/impl/halCommon.cpp:224-235)
extern const uint8_t dnaPackMap[256];
extern const char dnaUnpackMap[16];
inline char dnaAt(const std::vector<uint8_t> &packed, int64_t i) {
    uint8_t b = packed[(size_t)(i >> 1)];
    return dnaUnpackMap[(i & 1) ? (b & 0x0F) : (b >> 4)];
}
void packDna(const std::string &s, std::vector<uint8_t> &out);

// Synthetic alignment generator with halRandGen's semantics (randgen/halRandGen.cpp,
// api/tests/halRandomData.cpp).  Same seed + options => same trees, tilings, links and DNA.
struct RandOptions {
    double meanDegree = 1.25, maxBranchLength = 0.7;
    uint64_t minGenomes = 8, maxGenomes = 20, minSegmentLength = 500, maxSegmentLength = 2000, minSegments = 100,
             maxSegments = 500;
    int seed = -1;
    int withDna = 1; // 0: skip DNA content (not seed-compatible with halRandGen; benchmark use); 2: the alignment of 0 with
                     // DNA from a separate fast generator (hgx_randgen.cpp: FastDna)
};
bool randPreset(const std::string &name, RandOptions &opt);
Image createRandomAlignment(const RandOptions &opt);

} // namespace hgx
