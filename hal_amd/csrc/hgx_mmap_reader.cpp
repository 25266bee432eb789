// Read-only importer for mmap-format HAL files: walks the raw structs of
// /root/reference/api/mmap_impl and transposes them into the flat image (hgx_image.hpp).
//   header        mmapFile.h:23-31 (format[32] "HAL-MMAP", mmapVersion[32], halVersion[32], nextOffset@96,
//                 rootOffset@104, dirty@112); major version must be 1 (mmapFile.cpp:69-72); a dirty file is
//                 refused (mmapFile.cpp:96-98)
//   root object   mmapAlignment.h:14-31 (numGenomes, newick offset/length, genome array offset, name hash)
//   genome        mmapGenome.h:19-46 (12 x u64), name = MMapArray<char> (mmapArray.h:6-11: 24-byte header)
//   sequence      mmapSequenceData.h:20-30; 72 bytes in API 1.0 files, 328 bytes since API 1.1 (256 reserved)
//   top segment   mmapTopSegmentData.h:40-44, 40-byte records, numTop+1 of them (sentinel, mmapGenome.cpp:141)
//   bottom seg.   mmapBottomSegmentData.h:35-52, 8*(2+nc)+nc bytes rounded up to 8
//   DNA           mmapGenome.cpp:41-43, (len+1)/2 nibble-packed bytes
// The perfect hash (names) and the site-map BST are not read: names are compared directly and
// site->sequence is a binary search on the sequence start table.
#include "hgx_image.hpp"
#include <cstring>
#include <fcntl.h>
#include <map>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace hgx {

namespace {

struct Mapping {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    explicit Mapping(const std::string &path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0)
            throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd, &st) < 0) {
            ::close(fd);
            throw std::runtime_error("cannot stat " + path);
        }
        n = (size_t)st.st_size;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            throw std::runtime_error("cannot mmap " + path);
        }
        p = (const uint8_t *)m;
    }
    ~Mapping() {
        if (p)
            munmap((void *)p, n);
        if (fd >= 0)
            ::close(fd);
    }
    const uint8_t *at(uint64_t off, uint64_t len) const {
        if (off > n || len > n - off)
            throw std::runtime_error("mmap HAL: offset out of bounds, probably file corruption");
        return p + off;
    }
    uint64_t u64(uint64_t off) const {
        uint64_t v;
        memcpy(&v, at(off, 8), 8);
        return v;
    }
    int64_t s64(uint64_t off) const {
        return (int64_t)u64(off);
    }
};

} // namespace

Image readMmapHal(const std::string &path) {
    Mapping m(path);
    if (m.n < 120)
        throw std::runtime_error(path + ": file too small for an mmap HAL header");
    if (strncmp((const char *)m.at(0, 32), "HAL-MMAP", 32) != 0)
        throw std::runtime_error(path + ": invalid file header, expected format name of 'HAL-MMAP'");
    std::string ver((const char *)m.at(32, 32), strnlen((const char *)m.at(32, 32), 32));
    size_t dot = ver.find('.');
    if (dot == std::string::npos)
        throw std::runtime_error(path + ": doesn't have a valid mmap version string: " + ver.substr(0, 20));
    int major = atoi(ver.substr(0, dot).c_str()), minor = atoi(ver.substr(dot + 1).c_str());
    if (major != 1)
        throw std::runtime_error(path + ": incompatible mmap major versions: file version " + ver + ", mmap API version 1.1");
    uint64_t nextOffset = m.u64(96), rootOffset = m.u64(104);
    if (nextOffset > m.n || rootOffset > m.n)
        throw std::runtime_error(path + ": header offset field out of bounds, probably file corruption");
    if (*m.at(112, 1))
        throw std::runtime_error(path + ": file is marked as dirty, most likely an inconsistent state.");
    const uint64_t seqStride = (minor >= 1) ? 328 : 72;

    uint64_t numGenomes = m.u64(rootOffset), newickOff = m.u64(rootOffset + 8), newickLen = m.u64(rootOffset + 16),
             genomeArr = m.u64(rootOffset + 24);
    Image img;
    {
        const char *nw = (const char *)m.at(newickOff, newickLen);
        img.newick.assign(nw, strnlen(nw, newickLen));
    }
    NewickParser np(img.newick);
    int rootNode = np.parse();
    (void)rootNode;

    // genome records, in file (array) order = genome id
    img.genomes.resize(numGenomes);
    std::map<std::string, int> byName;
    std::vector<uint64_t> topOff(numGenomes), botOff(numGenomes);
    for (uint64_t g = 0; g < numGenomes; ++g) {
        uint64_t rec = genomeArr + 96 * g;
        GenomeTables &G = img.genomes[g];
        G.totalLength = m.s64(rec);
        uint64_t numSeq = m.u64(rec + 8);
        G.numTop = m.s64(rec + 16);
        G.numBot = m.s64(rec + 24);
        uint64_t nameOff = m.u64(rec + 32), seqOff = m.u64(rec + 56), dnaOff = m.u64(rec + 72);
        topOff[g] = m.u64(rec + 80);
        botOff[g] = m.u64(rec + 88);
        uint64_t nameLen = m.u64(nameOff + 16);
        const char *nm = (const char *)m.at(nameOff + 24, nameLen);
        G.name.assign(nm, strnlen(nm, nameLen));
        byName[G.name] = (int)g;
        G.seqs.resize(numSeq);
        for (uint64_t s = 0; s < numSeq; ++s) {
            uint64_t so = seqOff + seqStride * s;
            SeqInfo &S = G.seqs[s];
            S.start = m.s64(so);
            S.length = m.s64(so + 16);
            S.topStart = m.s64(so + 24);
            S.botStart = m.s64(so + 32);
            S.numTop = m.s64(so + 40);
            S.numBot = m.s64(so + 48);
            uint64_t nl = m.u64(so + 56), no = m.u64(so + 64);
            const char *sn = (const char *)m.at(no, nl);
            S.name.assign(sn, strnlen(sn, nl));
        }
        uint64_t dnaBytes = ((uint64_t)G.totalLength + 1) / 2;
        const uint8_t *dp = m.at(dnaOff, dnaBytes);
        G.dna.assign(dp, dp + dnaBytes);
    }
    // tree topology from the Newick text; child slot k = k-th child in the text (mmapAlignment.h:145-153)
    for (const NewickNode &nd : np.nodes) {
        auto it = byName.find(nd.label);
        if (it == byName.end())
            throw std::runtime_error(path + ": genome '" + nd.label + "' of the tree has no genome record");
        GenomeTables &G = img.genomes[(size_t)it->second];
        for (int k : nd.kids) {
            auto ck = byName.find(np.nodes[(size_t)k].label);
            if (ck == byName.end())
                throw std::runtime_error(path + ": genome '" + np.nodes[(size_t)k].label + "' of the tree has no genome record");
            G.children.push_back(ck->second);
            img.genomes[(size_t)ck->second].parent = it->second;
            img.genomes[(size_t)ck->second].branchLength = np.nodes[(size_t)k].len;
        }
    }
    // segment tables
    forEachGenome((size_t)numGenomes, [&](size_t g) {
        GenomeTables &G = img.genomes[g];
        const uint64_t nt = (uint64_t)G.numTop, nb = (uint64_t)G.numBot, nc = G.children.size();
        G.tStart.resize(nt + 1);
        G.tParent.resize(nt);
        G.tParalogy.resize(nt);
        G.tBotParse.resize(nt);
        G.tParentRev.resize(nt);
        if (nt > 0) {
            const uint8_t *tp = m.at(topOff[g], 40 * (nt + 1));
            for (uint64_t i = 0; i <= nt; ++i) {
                int64_t rec[4];
                memcpy(rec, tp + 40 * i, 32);
                G.tStart[i] = rec[0];
                if (i < nt) {
                    G.tBotParse[i] = rec[1];
                    G.tParalogy[i] = rec[2];
                    G.tParent[i] = rec[3];
                    G.tParentRev[i] = tp[40 * i + 32] ? 1 : 0;
                }
            }
        } else {
            G.tStart[0] = G.totalLength;
        }
        const uint64_t bsz = 8 * (2 + nc) + nc + ((nc % 8) ? 8 - nc % 8 : 0);
        G.bStart.resize(nb + 1);
        G.bTopParse.resize(nb);
        G.bChild.assign(nc, std::vector<int64_t>(nb));
        G.bChildRev.assign(nc, std::vector<uint8_t>(nb));
        if (nb > 0) {
            const uint8_t *bp = m.at(botOff[g], bsz * (nb + 1));
            for (uint64_t i = 0; i <= nb; ++i) {
                const uint8_t *r = bp + bsz * i;
                int64_t v;
                memcpy(&v, r, 8);
                G.bStart[i] = v;
                if (i < nb) {
                    memcpy(&v, r + 8, 8);
                    G.bTopParse[i] = v;
                    for (uint64_t k = 0; k < nc; ++k) {
                        memcpy(&v, r + 16 + 8 * k, 8);
                        G.bChild[k][i] = v;
                        G.bChildRev[k][i] = r[16 + 8 * nc + k] ? 1 : 0;
                    }
                }
            }
        } else {
            G.bStart[0] = G.totalLength;
        }
        // the sentinel records carry the end coordinate; be tolerant of files where it was left unset
        if (nt > 0 && G.tStart[nt] != G.totalLength)
            G.tStart[nt] = G.totalLength;
        if (nb > 0 && G.bStart[nb] != G.totalLength)
            G.bStart[nb] = G.totalLength;
    });
    return img;
}

} // namespace hgx
