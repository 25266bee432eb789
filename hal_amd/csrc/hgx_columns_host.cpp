#include "hgx_columns_host.hpp"
#include <functional>
#include "hgx_liftover_host.hpp"
#include "hgx_wig_text.hpp"
#include "hgx_textmem.hpp"
#include <iostream>
#include <algorithm>
#include <atomic>
#include <sstream>
#include <chrono>
#include <ctime>
#include <climits>
#include <cstdlib>
#include <charconv>
#include <climits>
#include <thread>
#include <cstring>
#include <condition_variable>
#include <mutex>

namespace hgx {

// ---------------------------------------------------------------------------------------------
// halAlignmentDepth
static void depthSequence(std::ostream &os, hgx_alignment *h, int genome, int seqIdx, const std::set<int> &targetSet, int64_t start,
                          int64_t length, int64_t step, bool countDupes, bool noAncestors, ColumnStats *stats,
                          const std::vector<hgx_alignment *> *moreDevices) {
    // printSequence, alignmentDepth/halAlignmentDepth.cpp:215-308
    const GenomeTables &G = h->img.genomes[(size_t)genome];
    const SeqInfo &S = G.seqs[(size_t)seqIdx];
    const int64_t seqLen = S.length;
    if (seqLen == 0)
        return;
    if (length == 0)
        length = seqLen - start;
    const int64_t last = start + length;
    if (last > seqLen)
        throw std::runtime_error("Specified range [" + std::to_string(start) + "," + std::to_string(length) + "] is" +
                                 "out of range for sequence " + S.name + ", which has length " + std::to_string(seqLen));
    if (step < 1)
        throw std::runtime_error("step must be positive");
    os << "fixedStep chrom=" << S.name << " start=" << start + 1 << " step=" << step << "\n";
    // positions the reference loop visits (:243-307): start, start+step, ... while inside [start, last); with step > 1 the
    // iterator is re-seated by toSite(pos, last) whose last column is `last` itself, so a position landing exactly on
    // `last` is also printed when it exists in the genome
    int64_t count = 0;
    if (step == 1) {
        count = length;
    } else if (length >= 1) {
        count = 1;
        if (length > 1) {
            int64_t p = start + step;
            while (p <= last && p + S.start < G.totalLength) {
                ++count;
                if (p >= last)
                    break;
                p += step;
            }
        }
    }
    ColumnOptions opt;
    opt.noAncestors = noAncestors;
    opt.targets.assign(targetSet.begin(), targetSet.end());
    // the lines: sizes counted and lines written by many threads, into the output itself where the stream gives room (hgx_wig_text.hpp)
    BulkSink *sink = dynamic_cast<BulkSink *>(os.rdbuf());
    auto lines = [&os, sink](const int32_t *v, int64_t n) { wigLines(os, v, n, [sink](size_t b) { return sink ? sink->room(b) : nullptr; }); };
    if (!(moreDevices && !moreDevices->empty() && count >= 2)) {
        // one device: the values come chunk by chunk (sixteen million of them) while the chunk before is made into lines
        // (HGX_WIG_CHUNK: columns per chunk — the tests run genomes of a few thousand columns through several chunks)
        int64_t chunk = (int64_t)1 << 24;
        if (const char *e = getenv("HGX_WIG_CHUNK"))
            chunk = std::max<int64_t>(1, atoll(e));
        // the lines made on the device where there is one (k_wig_text; HGX_WIG_DEVICE_TEXT=0: the host's threads): a chunk's text comes
        // back instead of its values, and is copied into the output by a few threads while the next chunk's is made
        if (h->dev && !(getenv("HGX_WIG_DEVICE_TEXT") && getenv("HGX_WIG_DEVICE_TEXT")[0] == '0')) {
            columnsDepthTextChunksHost(h, genome, start + S.start, count, step, countDupes ? 1 : 0, opt, stats, chunk, [&os, sink](const char *text, size_t bytes) {
                char *dst = sink && bytes >= ((size_t)1 << 20) ? sink->room(bytes) : nullptr;
                if (!dst) {
                    os.write(text, (std::streamsize)bytes);
                    return;
                }
                const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(hostThreads(), 8u), bytes >> 22));
                std::vector<std::thread> th;
                auto copy = [&](unsigned t) { memcpy(dst + bytes * t / parts, text + bytes * t / parts, bytes * (t + 1) / parts - bytes * t / parts); };
                for (unsigned t = 1; t < parts; ++t)
                    th.emplace_back(copy, t);
                copy(0);
                for (std::thread &x : th)
                    x.join();
            });
            return;
        }
        columnsDepthChunksHost(h, genome, start + S.start, count, step, countDupes ? 1 : 0, opt, stats, chunk,
                               [&lines](const int32_t *v, int64_t, int64_t n) { lines(v, n); });
        return;
    }
    // (a block the copies fill — hostBlockTake — not a vector cleared first)
    struct Values {
        int32_t *p;
        explicit Values(int64_t n) : p(static_cast<int32_t *>(hostBlockTake((size_t)std::max<int64_t>(n, 1) * 4))) {}
        ~Values() { hostBlockGive(p); }
        int32_t *data() const { return p; }
    } vals(count);
    {
        // columns are independent (api/impl/halColumnIterator.cpp:785-787): contiguous shares of the sampled columns, one per
        // device clone, scanned at the same time; the values land in their place of the one array
        std::vector<hgx_alignment *> hs{h};
        hs.insert(hs.end(), moreDevices->begin(), moreDevices->end());
        const int64_t nd = (int64_t)hs.size();
        std::vector<std::string> errors((size_t)nd);
        std::vector<std::thread> pool;
        for (int64_t d = 0; d < nd; ++d) {
            const int64_t lo = count * d / nd, hi = count * (d + 1) / nd;
            if (hi <= lo)
                continue;
            pool.emplace_back([&, d, lo, hi]() {
                try {
                    columnsDepthHost(hs[(size_t)d], genome, start + S.start + lo * step, hi - lo, step, countDupes ? 1 : 0, opt, vals.data() + lo,
                                     nullptr);
                } catch (std::exception &e) {
                    errors[(size_t)d] = e.what();
                }
            });
        }
        for (std::thread &t : pool)
            t.join();
        for (const std::string &e : errors)
            if (!e.empty())
                throw std::runtime_error(e);
    }
    lines(vals.data(), count);
}

void alignmentDepth(std::ostream &os, hgx_alignment *h, int genome, int sequence, const std::set<int> &targetSet, int64_t start,
                    int64_t length, int64_t step, bool countDupes, bool noAncestors, ColumnStats *stats,
                    const std::vector<hgx_alignment *> *moreDevices) {
    // printGenome, :318-347
    const GenomeTables &G = h->img.genomes[(size_t)genome];
    if (sequence >= 0) {
        depthSequence(os, h, genome, sequence, targetSet, start, length, step, countDupes, noAncestors, stats, moreDevices);
        return;
    }
    if (start + length > G.totalLength)
        throw std::runtime_error("Specified range [" + std::to_string(start) + "," + std::to_string(length) + "] is" +
                                 "out of range for genome " + G.name + ", which has length " + std::to_string(G.totalLength));
    if (length == 0)
        length = G.totalLength - start;
    int64_t runningLength = 0;
    for (size_t s = 0; s < G.seqs.size(); ++s) {
        const SeqInfo &S = G.seqs[s];
        if (start + length >= S.start && start < S.start + S.length && runningLength < length) {
            const int64_t readStart = S.start >= start ? 0 : start - S.start;
            int64_t readLen = std::min(S.length - readStart, length);
            readLen = std::min(readLen, length - runningLength);
            depthSequence(os, h, genome, (int)s, targetSet, readStart, readLen, step, countDupes, noAncestors, stats, moreDevices);
            runningLength += readLen;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// hal2maf
MafExport::~MafExport() {
    for (auto &kv : _entries)
        delete kv.second;
    delete _tree;
}

void MafExport::buildRanks() {
    // ColumnIterator::SequenceLess (api/inc/halColumnIterator.h:45-50): genome name, then sequence index
    const Image &img = _al->img;
    std::vector<int> order(img.genomes.size());
    for (size_t i = 0; i < order.size(); ++i)
        order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return img.genomes[(size_t)a].name < img.genomes[(size_t)b].name; });
    _rank.assign(img.genomes.size(), std::vector<int>());
    _rankGenome.clear();
    _rankSeq.clear();
    int r = 0;
    for (int g : order) {
        _rank[(size_t)g].resize(img.genomes[(size_t)g].seqs.size());
        for (size_t s = 0; s < img.genomes[(size_t)g].seqs.size(); ++s) {
            _rank[(size_t)g][s] = r++;
            _rankGenome.push_back(g);
            _rankSeq.push_back((int)s);
        }
    }
}

MafExport::Key MafExport::keyOf(int genome, int64_t pos) const {
    const GenomeTables &G = _al->img.genomes[(size_t)genome];
    const int s = G.seqs.size() == 1 ? 0 : G.seqIndexBySite(pos);
    return Key{_rank[(size_t)genome][(size_t)s], genome, s};
}

// halMafBlock.cpp:36-82
void MafExport::resetEntries() {
    _reference = nullptr;
    _refIndex = NULL_INDEX;
    for (auto i = _entries.begin(); i != _entries.end();) {
        Entry *e = i->second;
        if (e->start == NULL_INDEX) {
            if (e->lastUsed > 10) { // unused for more than 10 consecutive blocks: dropped
                delete e;
                i = _entries.erase(i);
                continue;
            }
            ++e->lastUsed;
        } else {
            e->lastUsed = 0;
        }
        e->start = NULL_INDEX;
        e->strand = '+';
        e->length = 0;
        e->sequence.clear();
        e->segs.clear();
        ++i;
    }
}

// halMafBlock.cpp:84-112
void MafExport::initEntry(Entry *e, const Key &k, const ColumnRowHost *row, bool clearSequence) {
    const GenomeTables &G = _al->img.genomes[(size_t)k.genome];
    const SeqInfo &S = G.seqs[(size_t)k.seq];
    if (e->genome != k.genome || e->srcLength != S.length || e->name.empty()) {
        e->name = _ucscNames ? G.name + "." + S.name : S.name; // Sequence::getFullName / getName (halMafBlock.h:128-130)
        auto it = _nameIds.find(e->name);
        if (it == _nameIds.end()) {
            it = _nameIds.emplace(e->name, (uint32_t)_names.size()).first;
            _names.push_back(e->name);
        }
        e->nameId = it->second;
        e->genome = k.genome;
        e->srcLength = S.length;
    }
    if (row) {
        e->start = row->pos - S.start;
        e->length = 0;
        e->strand = row->rev ? '-' : '+';
        if (row->rev)
            e->start = e->srcLength - 1 - e->start;
    } else {
        e->start = NULL_INDEX;
        e->length = 0;
        e->strand = '+';
    }
    if (clearSequence) {
        e->sequence.clear();
        e->segs.clear();
    }
    e->tree = nullptr;
}

// ---- hal2maf --printTree: the tree of a column's bases (halMafBlock.cpp:121-292), walked over the host's segment tables — a
// handful of hops per block, on the path that goes column by column anyway ----
namespace {
int64_t segmentAt(const std::vector<int64_t> &starts, int64_t n, int64_t pos) { // the segment that holds pos
    return (int64_t)(std::upper_bound(starts.begin(), starts.begin() + n, pos) - starts.begin()) - 1;
}
} // namespace
// :159-200: a node for a base: the entry of its sequence that continues there (or has no start yet), or, for a sequence the
// block has no entry of (an ancestor left out by --noAncestors), the genome's name
MafExport::Tree *MafExport::getTreeNode(int genome, int64_t pos, bool modifyEntries) {
    std::unique_ptr<Tree> ret(new Tree);
    const Key key = keyOf(genome, pos);
    const GenomeTables &G = _al->img.genomes[(size_t)genome];
    auto it = _entries.lower_bound(key);
    if (it != _entries.end() && it->first.rank == key.rank) {
        Entry *entry = nullptr;
        for (; it != _entries.end() && it->first.rank == key.rank; ++it) {
            Entry *cur = it->second;
            int64_t curPos = cur->start + cur->length;
            if (cur->strand == '-')
                curPos = cur->srcLength - 1 - curPos;
            if (curPos == pos - G.seqs[(size_t)key.seq].start || cur->start == NULL_INDEX) {
                entry = cur;
                break;
            }
        }
        if (!entry) // (an assertion in the reference: the column was found appendable before its tree is built)
            throw std::runtime_error("printTree: no block entry continues at this base");
        ret->entry = entry;
        ret->label = entry->name;
        if (modifyEntries)
            entry->tree = ret.get();
    } else {
        ret->label = G.name;
    }
    return ret.release();
}
// :204-237: under the node of a bottom segment's base, a node for the base of every child segment and of its paralogs, and on
// down through their parse links
void MafExport::buildTreeR(int genome, int64_t b, int64_t pos, Tree *node, bool modifyEntries) {
    const Image &img = _al->img;
    const GenomeTables &G = img.genomes[(size_t)genome];
    const int64_t off = pos - G.bStart[(size_t)b], len = G.bStart[(size_t)b + 1] - G.bStart[(size_t)b];
    for (size_t i = 0; i < G.children.size(); ++i) {
        const int64_t tc = G.bChild[i][(size_t)b];
        if (tc == NULL_INDEX)
            continue;
        const int c = G.children[i];
        const GenomeTables &CG = img.genomes[(size_t)c];
        int64_t t = tc;
        do { // the canonical paralog (the one the parent's link names) first, then the rest of the cycle
            const int64_t cpos = CG.tStart[(size_t)t] + (CG.tParentRev[(size_t)t] ? len - 1 - off : off);
            Tree *child = getTreeNode(c, cpos, modifyEntries);
            child->parent = node;
            node->children.push_back(child);
            if (CG.tBotParse[(size_t)t] != NULL_INDEX)
                buildTreeR(c, segmentAt(CG.bStart, CG.numBot, cpos), cpos, child, modifyEntries);
            t = CG.tParalogy[(size_t)t];
        } while (t != NULL_INDEX && t != tc);
    }
}
// :239-292: from the column's first base up to the segment that is the ancestor of all of them, then down
MafExport::Tree *MafExport::buildTree(const ColumnMap &col, bool modifyEntries) {
    const ColumnRowHost *first = nullptr;
    for (auto c = col.begin(); c != col.end() && !first; ++c)
        if (!c->second.empty())
            first = c->second[0];
    if (!first)
        throw std::runtime_error("printTree: empty column");
    const Image &img = _al->img;
    int g = first->genome;
    int64_t pos = first->pos;
    const GenomeTables &G0 = img.genomes[(size_t)g];
    if (G0.numTop == 0) { // the root genome
        std::unique_ptr<Tree> t(getTreeNode(g, pos, modifyEntries));
        buildTreeR(g, segmentAt(G0.bStart, G0.numBot, pos), pos, t.get(), modifyEntries);
        return t.release();
    }
    int64_t t = segmentAt(G0.tStart, G0.numTop, pos);
    int botGenome = -1;
    int64_t botPos = 0;
    while (img.genomes[(size_t)g].tParent[(size_t)t] != NULL_INDEX) {
        const GenomeTables &G = img.genomes[(size_t)g];
        const GenomeTables &P = img.genomes[(size_t)G.parent];
        const int64_t b = G.tParent[(size_t)t], off = pos - G.tStart[(size_t)t], len = G.tStart[(size_t)t + 1] - G.tStart[(size_t)t];
        botGenome = G.parent;
        botPos = P.bStart[(size_t)b] + (G.tParentRev[(size_t)t] ? len - 1 - off : off);
        if (P.parent < 0 || P.bTopParse[(size_t)b] == NULL_INDEX)
            break; // the root genome, or nothing above this segment
        g = botGenome;
        pos = botPos;
        t = segmentAt(P.tStart, P.numTop, pos);
    }
    const GenomeTables &G = img.genomes[(size_t)g];
    if (G.tParent[(size_t)t] == NULL_INDEX && g == first->genome && G.numBot == 0)
        return getTreeNode(g, pos, modifyEntries); // an insertion in a leaf: no bottom segment anywhere
    // (the reference dereferences a null bottom iterator here when the column's first base is an insertion in a genome that has
    // bottom segments: undefined there, an error here)
    if (botGenome < 0)
        throw std::runtime_error("printTree: the column's first base has no parent in a genome with bottom segments");
    const GenomeTables &B = img.genomes[(size_t)botGenome];
    std::unique_ptr<Tree> root(getTreeNode(botGenome, botPos, modifyEntries));
    buildTreeR(botGenome, segmentAt(B.bStart, B.numBot, botPos), botPos, root.get(), modifyEntries);
    return root.release();
}
namespace {
template <typename T> bool treeEquals(const T *a, const T *b) { // stTree_equals
    if (a->label != b->label || a->children.size() != b->children.size())
        return false;
    for (size_t i = 0; i < a->children.size(); ++i)
        if (!treeEquals(a->children[i], b->children[i]))
            return false;
    return true;
}
template <typename T> void treeNewick(const T *t, std::string &s) { // stTree_getNewickTreeString without the final ';'
    if (!t->children.empty()) {
        s += '(';
        for (size_t i = 0; i < t->children.size(); ++i) {
            if (i)
                s += ',';
            treeNewick(t->children[i], s);
        }
        s += ')';
    }
    s += t->label;
}
template <typename T> void prioritizeNodeInTree(T *node) { // :128-157: the node and its ancestors first among their siblings
    T *parent = node->parent;
    if (!parent)
        return;
    size_t at = 0;
    while (parent->children[at] != node)
        ++at;
    std::swap(parent->children[0], parent->children[at]);
    prioritizeNodeInTree(parent);
}
} // namespace

// halMafBlock.cpp:114-138
void MafExport::updateEntry(Entry *e, const Key *k, const ColumnRowHost *row) {
    if (row) {
        if (e->start == NULL_INDEX)
            initEntry(e, *k, row, false);
        ++e->length;
        if (_segMode)
            appendRun(e, row, row->pos, 1);
        else
            e->sequence.push_back(row->base);
    } else if (_segMode) {
        appendRun(e, nullptr, 0, 1);
    } else {
        e->sequence.push_back('-');
    }
}

// run mode: n more columns of this row starting at genome position pos (a gap when row is null); contiguous runs merge
void MafExport::appendRun(Entry *e, const ColumnRowHost *row, int64_t pos, int64_t n) {
    const uint8_t kind = !row ? 0 : (row->rev ? 2 : 1);
    if (!e->segs.empty()) {
        Entry::Seg &l = e->segs.back();
        if (l.kind == kind && (int64_t)l.n + n < INT32_MAX &&
            (kind == 0 || (kind == 1 ? pos == l.pos + l.n : pos == l.pos - l.n))) {
            l.n += (int32_t)n;
            return;
        }
    }
    while (n > 0) { // a run longer than 2^31 - 1 columns is split (not reachable with a block length limit, kept for safety)
        const int64_t m = std::min<int64_t>(n, INT32_MAX - 1);
        e->segs.push_back(Entry::Seg{pos, (int32_t)m, kind});
        pos += kind == 2 ? -m : m;
        n -= m;
    }
}

// halMafBlock.cpp:294-367
void MafExport::initBlock(const ColumnMap &col, const Key &refKey, int64_t refPos) {
    if (_printTree && _tree) {
        delete _tree;
        _tree = nullptr;
    }
    resetEntries();
    Entries::iterator e = _entries.begin();
    for (auto c = col.begin(); c != col.end(); ++c) {
        const Key &k = c->first;
        if (c->second.empty()) {
            e = _entries.lower_bound(k);
            if (e == _entries.end() || e->first.rank != k.rank) {
                Entry *ne = new Entry;
                initEntry(ne, k, nullptr);
                e = _entries.insert(Entries::value_type(k, ne));
            } else {
                initEntry(e->second, k, nullptr);
            }
        } else {
            for (const ColumnRowHost *row : c->second) {
                if (e == _entries.begin()) {
                    e = _entries.lower_bound(k);
                    if (e == _entries.end() || e->first.rank != k.rank)
                        e = _entries.end();
                } else {
                    while (e != _entries.end() && e->first.rank != k.rank)
                        ++e;
                }
                if (e == _entries.end()) {
                    Entry *ne = new Entry;
                    initEntry(ne, k, row);
                    e = _entries.insert(Entries::value_type(k, ne));
                } else {
                    initEntry(e->second, k, row);
                }
                ++e;
            }
        }
    }
    if (_reference == nullptr) {
        e = _entries.lower_bound(refKey);
        if (e == _entries.end() || e->first.rank != refKey.rank)
            e = _entries.begin();
        _reference = e->second;
        if (e->first.rank == refKey.rank)
            _refIndex = refPos;
    }
    if (_printTree)
        _tree = buildTree(col, true);
}

// halMafBlock.cpp:370-395
void MafExport::appendColumn(const ColumnMap &col) {
    Entries::iterator e = _entries.begin();
    for (auto c = col.begin(); c != col.end(); ++c) {
        for (const ColumnRowHost *row : c->second) {
            while (e != _entries.end() && e->first.rank != c->first.rank) {
                updateEntry(e->second, nullptr, nullptr);
                ++e;
            }
            updateEntry(e->second, &c->first, row);
            ++e;
        }
    }
    for (; e != _entries.end(); ++e)
        updateEntry(e->second, nullptr, nullptr);
}

// halMafBlock.cpp:401-452
bool MafExport::canAppendColumn(const ColumnMap &col) {
    Entries::iterator e = _entries.begin();
    for (auto c = col.begin(); c != col.end(); ++c) {
        if (c->second.empty())
            continue;
        const int64_t sequenceStart = _al->img.genomes[(size_t)c->first.genome].seqs[(size_t)c->first.seq].start;
        for (const ColumnRowHost *row : c->second) {
            while (e != _entries.end() && e->first.rank != c->first.rank)
                ++e;
            if (e == _entries.end())
                return false;
            const Entry *entry = e->second;
            if (entry->start != NULL_INDEX) {
                if (entry->length >= _maxBlockLength || (entry->length > 0 && (entry->strand == '-') != (row->rev != 0)))
                    return false;
                int64_t pos = row->pos - sequenceStart;
                if (row->rev)
                    pos = entry->srcLength - 1 - pos;
                if (pos - entry->start != entry->length)
                    return false;
            }
            ++e;
        }
    }
    if (_printTree) { // :443-448: the column's tree must be the block's
        std::unique_ptr<Tree> t(buildTree(col, false));
        return treeEquals(t.get(), _tree);
    }
    return true;
}

bool MafExport::referenceIsAllGaps() const {
    if (!_reference)
        return false;
    if (_segMode) {
        for (const Entry::Seg &g : _reference->segs)
            if (g.kind != 0)
                return false;
        return true;
    }
    for (char c : _reference->sequence)
        if (c != '-')
            return false;
    return true;
}

// halMafBlock.cpp:454-458, 499-519
namespace {
// packed byte -> its two bases as text; reversed: low nibble first (walking the reverse strand)
struct PairTable {
    uint16_t v[256];
    PairTable(const char *map, bool reversed) {
        for (int b = 0; b < 256; ++b) {
            const char hi = map[b >> 4], lo = map[b & 15];
            char two[2] = {reversed ? lo : hi, reversed ? hi : lo};
            memcpy(&v[b], two, 2);
        }
    }
};
} // namespace

// run mode: the rows of the block as data (MafBlock's operator<< order: the reference row first, then the entries that have a
// start, halMafBlock.cpp:499-520); text is made later, by flushSnapshots
void MafExport::snapshotBlock() {
    BlockSnap b{(uint32_t)_snapRows.size(), 0};
    auto add = [&](const Entry &e, int64_t start) {
        RowSnap r;
        r.nameId = e.nameId;
        r.firstSeg = (uint32_t)_snapSegs.size();
        r.numSegs = (uint32_t)e.segs.size();
        _snapSegs.insert(_snapSegs.end(), e.segs.begin(), e.segs.end());
        r.start = start;
        r.length = e.length;
        r.srcLength = e.srcLength;
        r.genome = e.genome;
        r.strand = e.strand;
        _snapRows.push_back(r);
        ++b.numRows;
    };
    if (_reference->start == NULL_INDEX) {
        if (_refIndex != NULL_INDEX)
            add(*_reference, _refIndex);
    } else {
        add(*_reference, _reference->start);
    }
    for (auto e = _entries.begin(); e != _entries.end(); ++e)
        if (e->second->start != NULL_INDEX && e->second != _reference)
            add(*e->second, e->second->start);
    _snapBlocks.push_back(b);
}

// renders the pending blocks ("a\n", the rows, one blank line each) on several threads and writes them in order
// — in the background: the batch is moved out, the state machine goes on filling the next one while this one is rendered
// and written (the previous batch is waited for first, so the stream sees the batches in order).
void MafExport::flushSnapshots(std::ostream &os) {
    waitPendingWrite();
    if (_snapBlocks.empty())
        return;
#ifdef HGX_HOST_PROFILE
    if (getenv("HGX_MAF_NO_RENDER")) { // (the state machine alone)
        _snapBlocks.clear();
        _snapRows.clear();
        _snapSegs.clear();
        return;
    }
#endif
    struct Batch {
        std::vector<BlockSnap> blocks;
        std::vector<RowSnap> rows;
        std::vector<Entry::Seg> segs;
    };
    auto batch = std::make_shared<Batch>();
    batch->blocks.swap(_snapBlocks);
    batch->rows.swap(_snapRows);
    batch->segs.swap(_snapSegs);
    // (the names by address, taken here: the state machine may add names while the batch is rendered; a deque's elements stay put)
    auto names = std::make_shared<std::vector<const std::string *>>();
    names->reserve(_names.size());
    for (const std::string &n : _names)
        names->push_back(&n);
    const hgx_alignment *al = _al;
    std::ostream *out = &os;
    _pendingWrite = std::async(std::launch::async, [batch, names, al, out]() {
    const std::vector<BlockSnap> &_snapBlocks = batch->blocks;
    const std::vector<RowSnap> &_snapRows = batch->rows;
    const std::vector<Entry::Seg> &_snapSegs = batch->segs;
    const std::vector<const std::string *> &_names = *names;
    const hgx_alignment *_al = al;
    std::ostream &os = *out;
    static const char fwd[17] = "acgtn\0\0\0ACGTN\0\0\0";
    static const char rc[17] = "tgcan\0\0\0TGCAN\0\0\0";
    static const PairTable fwd2(fwd, false), rc2(rc, true);
    const size_t nb = _snapBlocks.size();
    unsigned nt = hostThreads();
    nt = std::max(1u, std::min(nt ? nt : 1u, 16u));
    if (nb < 256)
        nt = 1;
    std::vector<std::string> text(nt);
    auto render = [&](unsigned t) {
        std::string &buf = text[t];
        const size_t b0 = nb * t / nt, b1 = nb * (t + 1) / nt;
        size_t bytes = 0;
        for (size_t b = b0; b < b1; ++b)
            for (uint32_t k = 0; k < _snapBlocks[b].numRows; ++k) {
                const RowSnap &r = _snapRows[_snapBlocks[b].firstRow + k];
                bytes += _names[r.nameId]->size() + 80;
                for (uint32_t g = 0; g < r.numSegs; ++g)
                    bytes += (size_t)_snapSegs[r.firstSeg + g].n;
            }
        buf.resize(bytes + 3 * (b1 - b0));
        char *o = &buf[0];
        auto num = [&](int64_t v) { o = std::to_chars(o, o + 24, v).ptr; };
        for (size_t b = b0; b < b1; ++b) {
            *o++ = 'a';
            *o++ = '\n';
            for (uint32_t k = 0; k < _snapBlocks[b].numRows; ++k) {
                const RowSnap &r = _snapRows[_snapBlocks[b].firstRow + k];
                *o++ = 's';
                *o++ = '\t';
                const std::string &nm = *_names[r.nameId];
                memcpy(o, nm.data(), nm.size());
                o += nm.size();
                *o++ = '\t';
                num(r.start);
                *o++ = '\t';
                num(r.length);
                *o++ = '\t';
                *o++ = r.strand;
                *o++ = '\t';
                num(r.srcLength);
                *o++ = '\t';
                const std::vector<uint8_t> &d = _al->img.genomes[(size_t)r.genome].dna;
                const uint8_t *pk = d.data();
                for (uint32_t g = 0; g < r.numSegs; ++g) {
                    const Entry::Seg &sg = _snapSegs[r.firstSeg + g];
                    const int64_t t = sg.n;
                    if (sg.kind == 0) {
                        memset(o, '-', (size_t)t);
                    } else if (d.empty()) {
                        memset(o, 'N', (size_t)t);
                    } else if (sg.kind == 1) { // dnaUnpack (halCommon.h:187-190), two bases per packed byte
                        int64_t p0 = sg.pos, i = 0;
                        if (i < t && (p0 & 1)) {
                            o[i++] = fwd[pk[p0 >> 1] & 0x0F];
                            ++p0;
                        }
                        for (; i + 1 < t; i += 2, p0 += 2)
                            memcpy(o + i, &fwd2.v[pk[p0 >> 1]], 2);
                        if (i < t)
                            o[i] = fwd[pk[p0 >> 1] >> 4];
                    } else { // reverse strand: walk left, complemented (reverseComplement, halCommon.h:45-75)
                        int64_t p0 = sg.pos, i = 0;
                        if (i < t && !(p0 & 1)) {
                            o[i++] = rc[pk[p0 >> 1] >> 4];
                            --p0;
                        }
                        for (; i + 1 < t; i += 2, p0 -= 2)
                            memcpy(o + i, &rc2.v[pk[p0 >> 1]], 2);
                        if (i < t)
                            o[i] = rc[pk[p0 >> 1] & 0x0F];
                    }
                    o += t;
                }
                *o++ = '\n';
            }
            *o++ = '\n';
        }
        buf.resize((size_t)(o - buf.data()));
    };
    if (nt == 1) {
        render(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back(render, t);
        for (std::thread &x : th)
            x.join();
    }
    for (const std::string &t : text)
        os.write(t.data(), (std::streamsize)t.size());
    });
}

void MafExport::printBlock(std::ostream &os) const {
    // MafBlock's operator<< (halMafBlock.cpp:499-520) prints field by field through the stream; formatting a block into
    // one buffer and writing it once is the same bytes at a fifth of the cost (the stream formatting was half of hal2maf's
    // run time here)
    static thread_local std::string buf;
    buf.clear();
    auto num = [&](int64_t v) {
        char tmp[24];
        auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
        buf.append(tmp, (size_t)(r.ptr - tmp));
    };
    auto printEntry = [&](const Entry &e, int64_t start) {
        buf += "s\t";
        buf += e.name;
        buf += '\t';
        num(start);
        buf += '\t';
        num(e.length);
        buf += '\t';
        buf += e.strand;
        buf += '\t';
        num(e.srcLength);
        buf += '\t';
        buf += e.sequence;
        buf += '\n';
    };
    if (_printTree) { // printBlockWithTree (halMafBlock.cpp:485-497): the tree as a comment of the block, the rows in post order
        if (_reference->tree)
            prioritizeNodeInTree(_reference->tree); // the reference first
        buf += "a tree=\"";
        treeNewick(_tree, buf);
        buf += ";\"\n";
        std::function<void(const Tree *)> rows = [&](const Tree *t) {
            for (const Tree *c : t->children)
                rows(c);
            if (t->entry) // (none for an ancestor left out by --noAncestors)
                printEntry(*t->entry, t->entry->start);
        };
        rows(_tree);
        os.write(buf.data(), (std::streamsize)buf.size());
        return;
    }
    buf += "a\n";
    if (_reference->start == NULL_INDEX) {
        if (_refIndex != NULL_INDEX)
            printEntry(*_reference, _refIndex);
    } else {
        printEntry(*_reference, _reference->start);
    }
    for (auto e = _entries.begin(); e != _entries.end(); ++e)
        if (e->second->start != NULL_INDEX && e->second != _reference)
            printEntry(*e->second, e->second->start);
    os.write(buf.data(), (std::streamsize)buf.size());
}

// The default export path (no --unique, no --maxRefGap): run-compressed columns from the device, MafBlock's state machine
// (maf/impl/halMafBlock.cpp:36-82 resetEntries, :294-367 initBlock, :370-395 appendColumn, :401-450 canAppendColumn) on flat
// arrays sorted by the rank of the sequence instead of the reference's multimap / map of vectors.  What the reference's
// iterator-chasing loops amount to, stated once: the entries and the column's bases are both ordered by sequence; the i-th base
// of a sequence goes with the i-th entry of that sequence; initBlock makes the entries that are missing (behind the ones the
// sequence has), and gives every sequence the column map still has a key for (its bases are gone, the key stays until
// defragment) one entry at least; canAppendColumn fails when a sequence has more bases than entries.
//
// Only what decides where blocks begin is sequential: per entry its start, length, strand and age.  That is all the thread
// that walks the columns keeps; of every block it leaves a log — the entries it had, and per appended stretch of columns which
// base each entry was given — and the rows themselves (runs of gaps and bases, then text) are made from the log by the
// rendering threads, a batch of blocks at a time, beside the walk.
#ifdef HGX_HOST_PROFILE
// the profiling build's record of the device's batches, one file for both export paths, in the order of the calls
static FILE *mafReplayFile() {
    static FILE *f = getenv("HGX_MAF_REPLAY") ? fopen(getenv("HGX_MAF_REPLAY"), "rb") : nullptr;
    return f;
}
static FILE *mafDumpFile() {
    static FILE *f = getenv("HGX_MAF_DUMP") ? fopen(getenv("HGX_MAF_DUMP"), "wb") : nullptr;
    return f;
}
static unsigned long long g_mafTicks[12];
struct MafTick {
    int slot;
    unsigned long long t0;
    explicit MafTick(int s) : slot(s), t0(__builtin_ia32_rdtsc()) {}
    ~MafTick() { g_mafTicks[slot] += __builtin_ia32_rdtsc() - t0; }
};
#ifdef HGX_HOST_TICKS // (two rdtsc a scope: the walk's phases are a few hundred ticks a column)
#define MAF_TICK(slot) MafTick mafTick##slot(slot)
#else
#define MAF_TICK(slot)
#endif
#else
#define MAF_TICK(slot)
#endif
struct MafExport::RunMachine {
    struct PRow { // a base of a column: what the walk and the renderers need of it, sixteen bytes
        int64_t key;  // 2 * (its place in its sequence, counted on its strand: an entry's start) + (reverse strand)
        int32_t rank; // of its sequence (genome, sequence, the sequence's start and length: RankInfo)
        uint32_t ord; // its place in the column as the walk delivers it (the order of a sequence's bases in the column map)
    };
    struct Chunk { // one device batch: which columns are heads, the heads' rows (sorted the way the column map holds them)
        int64_t done = 0, n = 0;
        std::vector<uint8_t> head;
        std::vector<uint32_t> headOff;
        std::vector<uint32_t> headCol; // (the walk over slices: the column of every head, so that a walk can begin in the middle of a chunk)
        struct GiveBack {
            void operator()(PRow *p) const { hostBlockGive(p); }
        };
        std::unique_ptr<PRow[], GiveBack> rows; // (a block of the pool the device's copies use: touched before, not paged in again)
        double seconds = 0;
    };
    struct RankInfo {
        int64_t nameId = -1, srcLength = 0, seqStart = 0;
        int32_t genome = 0, seq = 0;
        // what every row of the sequence's begins with and what stands between its strand and its bases ("s\t<name>\t", "\t<length of
        // the sequence>\t": halMafBlock.cpp:499-520 prints them field by field), made once when the sequence is first met
        std::string rowHead, rowTail;
    };
    struct BlockLog {
        uint32_t firstEnt, numEnts, firstEvent, numEvents;
        int32_t refEnt;
        int64_t refIndex;
    };
    struct EventLog {
        int64_t k;       // columns
        const PRow *rows; // the bases of the event's first column, sorted the way the column map holds them
        uint32_t firstIdx, nRows; // rowEnt[firstIdx + r]: the entry of the block that base r was given to
    };
    struct Batch {
        std::vector<BlockLog> blocks;
        std::vector<int32_t> entRank;
        std::vector<EventLog> events;
        // (per event only which entry each BASE went to — a dozen words —, not a pointer per ENTRY of the block: the rendering
        // threads turn it round; the walk is the one thread everything waits for)
        std::vector<uint32_t> rowEnt;
        size_t numIdx = 0; // (rowEnt is grown ahead of the walk; its first numIdx words are the log)
        std::vector<std::shared_ptr<Chunk>> chunks; // (what the events' rows point into)
        std::vector<std::unique_ptr<PRow[]>> extra;
        void reset() { // (the arrays keep their memory)
            blocks.clear();
            entRank.clear();
            events.clear();
            numIdx = 0;
            chunks.clear();
            extra.clear();
        }
    };
    // A batch's log is tens of megabytes written once by the walk: taken fresh, every page of it is a fault in the one thread
    // everything waits for.  Rendered batches come back here and are written over.
    struct BatchPool {
        std::mutex mu;
        std::vector<std::unique_ptr<Batch>> idle;
    };
    static BatchPool &batchPool() {
        static BatchPool *pool = new BatchPool;
        return *pool;
    }
    // HGX_MAF_BATCH_POOL=1: rendered batches' logs come back and are written over.  Off by default: on the GPU box a log that
    // 32 rendering threads have just read lies in their caches, and the walk's stores into it wait for those lines one by one
    // (place(): 429 -> 154 Mticks per config-3 export with fresh logs, profiles/r04y_gpu_maf_diag_pool.txt); on a VM whose page
    // faults cost 2 us each the pool is the faster one.
    static bool poolBatches() {
        const char *e = getenv("HGX_MAF_BATCH_POOL");
        return e && atoi(e) != 0;
    }
    static std::unique_ptr<Batch> takeBatch() {
        BatchPool &pool = batchPool();
        if (poolBatches()) {
            std::lock_guard<std::mutex> lock(pool.mu);
            if (!pool.idle.empty()) {
                std::unique_ptr<Batch> b = std::move(pool.idle.back());
                pool.idle.pop_back();
                return b;
            }
        }
        return std::unique_ptr<Batch>(new Batch);
    }
    static void giveBatch(std::unique_ptr<Batch> b) {
        if (!poolBatches())
            return;
        b->reset();
        BatchPool &pool = batchPool();
        std::lock_guard<std::mutex> lock(pool.mu);
        if (pool.idle.size() < 3)
            pool.idle.push_back(std::move(b));
    }
    MafExport &M;
    std::ostream &os;
    const Image &img;
    int refRank; // the rank of the reference sequence of the column a block begins with (the column-by-column path sets it per column)
    // The block's entries (MafBlock::_entries), sorted by the rank of their sequence, as the four things the walk asks of them,
    // each in an array of its own.  Nothing is written into an entry when a block begins (resetEntries, halMafBlock.cpp:36-82,
    // visits every entry): an entry says which block gave it a base last, and what it holds — where its next base has to
    // be, how long it is — counts only while that is the block being made.  Its age, the reference's _lastUsed, is how far that
    // block lies back: "unused for more than ten blocks in a row" is block - last >= 13 at the beginning of a block.
    struct Ent {
        int64_t next; // the key (PRow::key) its next base must have: 2 * (start + length) + strand
        int64_t len;  // its length
        int64_t src;  // the length of its sequence
    };
    std::vector<int32_t> erank;
    std::vector<uint8_t> ekey;  // the entry was made for a key of the column map and has not been given a base since (the walk over slices)
    std::vector<int64_t> lastSeen; // per rank: the block (counted over the export) in which the sequence last had a base; INT64_MIN: not seen
    std::vector<int32_t> elast; // the block that gave the entry a base last (an entry made for a block that gave it none: the block before)
    std::vector<Ent> ent;
    std::vector<int32_t> firstOf;
    int32_t block = -1; // the block being made, counted from 0 (-1: the entries are the ones the constructor found)
    std::vector<int32_t> keys; // the column map's keys (ranks), the ones without bases in the current column among them
    std::vector<uint8_t> inKeys;
    std::shared_ptr<std::vector<RankInfo>> rankInfo;
    std::unique_ptr<Batch> batch;
    std::shared_ptr<Chunk> chunk;
    BlockLog cur{};
    size_t appendCount = 0, numBlocks = 0;
    bool entsLogged = false; // the entries' ranks as they are stand in this batch's log (at lastFirstEnt): the next block points there too
    uint32_t lastFirstEnt = 0;
    size_t refHint = 0;
    // ---- a walk over a SLICE of the export (convertSequenceRunsSliced) ----
    // What decides where blocks begin, and so the whole log from a column on, is: the entries (their sequences, what the block being
    // made has given them, how many blocks ago they were last given a base), the column map's keys, and the count of blocks so far
    // (defragment runs at every thousandth).  A walk that begins cold some thousand blocks in front of its slice has all of that
    // right when it reaches the slice — if it was told the right count.  It logs the blocks that BEGIN inside [logFrom, stopAt),
    // and what it looked like when its first and its one-past-last block began is kept: the slice before must have ended as this
    // one began (Snapshot equality), which makes this slice's log the sequential walk's by induction from the first slice.
    struct Snapshot {
        int64_t column = -1; // the reference position of the column the block begins with
        size_t numBlocks = 0;
        std::vector<int32_t> erank, age, keys;
        std::vector<int64_t> next, len; // (of entries the block being made has given bases: age 0)
        // what does not depend on the count of blocks the walk was told: the entries that bases made, and how many blocks ago every
        // sequence was last seen (up to a thousand: the keys are a function of that and of the count)
        std::vector<int32_t> bRank, bAge;
        std::vector<std::pair<int32_t, int32_t>> seen;
        bool operator==(const Snapshot &o) const {
            return column == o.column && numBlocks == o.numBlocks && erank == o.erank && age == o.age && keys == o.keys && next == o.next && len == o.len;
        }
        bool sameButForTheCount(const Snapshot &o) const {
            return column == o.column && bRank == o.bRank && bAge == o.bAge && next == o.next && len == o.len && seen == o.seen;
        }
    };
    // How a slice's own count of blocks moves with the count it is told.  The count matters through defragment alone (the column map's
    // keys are reset at the beginning of every block whose number is 1 above a multiple of a thousand), the keys matter through the
    // entries initBlock makes for keys without one, and those through pair() alone: a sequence that turns up in the middle of a block
    // finds such an entry (no break) or none (a break).  Every such look that decided a column is noted — the block it happened in, the
    // block the sequence was last seen in — and keyEntryAt says what it would have found under another count: the next round's walks
    // are told counts put right by that (countChangeUnderShift).  An ESTIMATE that saves rounds; what is accepted is decided by the
    // states alone (Snapshot equality, walkSliced).
    struct KeyUse {
        int64_t seen, at; // blocks counted over the export as this walk was told (seen: INT64_MIN = not since this walk began)
        bool found;       // the sequence had an entry (a key's)
    };
    std::vector<KeyUse> keyUses;
    int64_t firstBlock = 0; // the count this walk began with
    // When does a sequence last seen in block `seen` have an entry in block `at` although no base has made one (at >= seen + 13: the
    // entry its bases made is dropped at the beginning of block seen + 13)?  initBlock makes one at the beginning of a block for every key
    // of the column map without an entry; it lives twelve blocks and is made again while the sequence is still a key: made at c0 =
    // seen + 13, c0 + 12, c0 + 24, ... as long as no reset has fallen behind `seen` — the first block behind `seen` that begins with a
    // reset (block g begins with one when g - 1 is a multiple of a thousand) ends that; the last one made lives on for its twelve blocks.
    // shift: the blocks' true numbers are these plus shift.  creation: the block the entry in place at `at` was made in.
    static int64_t firstResetBehind(int64_t seen, int64_t shift) {
        int64_t y = seen + shift; // g + shift - 1 for g = seen + 1: the smallest multiple of a thousand at or above it (and 0)
        if (y < 0)
            y = 0;
        return (y + 999) / 1000 * 1000 - shift + 1;
    }
    static bool keyEntryAt(int64_t seen, int64_t at, int64_t shift, int64_t *creation = nullptr) {
        const int64_t c0 = seen + 13;
        if (at < c0)
            return false;
        const int64_t r1 = firstResetBehind(seen, shift);
        if (r1 <= c0)
            return false;
        const int64_t made = r1 > at ? c0 + (at - c0) / 12 * 12 : c0 + (r1 - 1 - c0) / 12 * 12;
        if (creation)
            *creation = made;
        return at <= made + 11;
    }
    // how many more blocks (fewer: negative) this walk would have counted had it been told a count `shift` higher, as far as the looks
    // at keys' entries tell: one that found an entry and would find none is a block more, the other way round one fewer (what follows
    // from that further on is left out: an estimate for the next round's counts, not a result)
    int64_t countChangeUnderShift(int64_t shift) const {
        int64_t change = 0;
        for (const KeyUse &u : keyUses) // (in the walk's order: a block more or fewer moves the numbers of the blocks behind it)
            if (u.seen != INT64_MIN && keyEntryAt(u.seen, u.at, shift + change) != u.found)
                change += u.found ? 1 : -1;
        return change;
    }
    bool irregularKeys = false; // keys that no base of a block made (--unique: columns walked and not written): the model above does not hold
    // everything the walk goes on from (not the log)
    struct MachineState {
        std::vector<int32_t> erank, elast, firstOf, keys;
        std::vector<Ent> ent;
        std::vector<uint8_t> inKeys, ekey;
        std::vector<int64_t> lastSeen;
        int32_t block = -1;
        size_t appendCount = 0, numBlocks = 0;
    };
    MachineState saveState() const {
        return MachineState{erank, elast, firstOf, keys, ent, inKeys, ekey, lastSeen, block, appendCount, numBlocks};
    }
    void restoreState(const MachineState &st) {
        erank = st.erank;
        elast = st.elast;
        firstOf = st.firstOf;
        keys = st.keys;
        ent = st.ent;
        inKeys = st.inKeys;
        ekey = st.ekey;
        lastSeen = st.lastSeen;
        block = st.block;
        appendCount = st.appendCount;
        numBlocks = st.numBlocks;
        refHint = 0;
        entsLogged = false;
    }
    // the state in front of the head in whose columns the next slice's first block begins (kept from the first head at or behind the
    // slice's end on): the next slice can be walked from it exactly, in the place of a cold run-up
    MachineState preStop;
    size_t preStopChunk = 0, preStopHead = 0, curChunkIndex = 0;
    bool preStopValid = false;
    bool owner = true;     // the constructor took the export's entries and the destructor hands them back
    bool sliced = false;   // logs are kept (not rendered) and cut at the slice's ends
    int64_t logFrom = INT64_MIN, stopAt = INT64_MAX;
    bool logging = true, stopped = false, startPending = false;
    Snapshot startSnap, endSnap;
    size_t warmBlocks = 0; // blocks begun in front of the slice (the cold walk's run-up)
    int64_t expectCount = -1; // >= 0: the count of blocks the slice's first block must begin at (else the walk stops there: miscounted)
    bool miscounted = false;
    bool emptySlice = false; // the first block at or behind the slice's beginning begins behind its end, at emptyAt
    int64_t emptyAt = -1;
    size_t numBlocksAtStart = 0;
    struct LogMark {
        size_t blocks, events, numIdx, entRank, extra;
    } stopMark{};
    Snapshot snapshot(int64_t column) const {
        Snapshot sn;
        sn.column = column;
        sn.numBlocks = numBlocks;
        sn.erank = erank;
        sn.keys = keys;
        sn.age.resize(erank.size());
        for (size_t i = 0; i < erank.size(); ++i) {
            sn.age[i] = block - elast[i];
            if (elast[i] == block) {
                sn.next.push_back(ent[i].next);
                sn.len.push_back(ent[i].len);
            }
            if (!ekey[i]) {
                sn.bRank.push_back(erank[i]);
                sn.bAge.push_back(block - elast[i]);
            }
        }
        for (size_t r = 0; r < lastSeen.size(); ++r)
            if (lastSeen[r] != INT64_MIN && (int64_t)numBlocks - lastSeen[r] <= 1000)
                sn.seen.emplace_back((int32_t)r, (int32_t)((int64_t)numBlocks - lastSeen[r]));
        return sn;
    }
    // a walk that owns nothing of the export: cold (no entries, no keys), told the count of blocks in front of it
    RunMachine(MafExport &m, std::ostream &o, int refRank_, std::shared_ptr<std::vector<RankInfo>> ranks, size_t numBlocks_)
        : M(m), os(o), img(m._al->img), refRank(refRank_), rankInfo(std::move(ranks)), batch(new Batch), numBlocks(numBlocks_), owner(false), sliced(true) {
        inKeys.assign(rankInfo->size(), 0);
        firstOf.assign(rankInfo->size(), -1);
        lastSeen.assign(rankInfo->size(), INT64_MIN);
        firstBlock = (int64_t)numBlocks_;
    }
    // every sequence's name and length looked up now: walks on several threads only read them
    void describeAllRanks() {
        for (size_t r = 0; r < rankInfo->size(); ++r)
            info((int32_t)r);
    }

    RunMachine(MafExport &m, std::ostream &o, int refRank_) : M(m), os(o), img(m._al->img), refRank(refRank_), batch(takeBatch()) {
        size_t ranks = 0;
        for (const std::vector<int> &r : M._rank)
            ranks += r.size();
        rankInfo = std::make_shared<std::vector<RankInfo>>(ranks);
        inKeys.assign(ranks, 0);
        firstOf.assign(ranks, -1);
        lastSeen.assign(ranks, INT64_MIN);
        for (auto &kv : M._entries) { // the block the other paths (and the sequence before) left
            const Entry &e = *kv.second;
            erank.push_back(kv.first.rank);
            const bool used = e.start != NULL_INDEX;
            ekey.push_back(0);
            elast.push_back(used ? -1 : -2 - (int32_t)e.lastUsed);
            ent.push_back(Ent{used ? ((e.start + e.length) << 1) | (e.strand == '-' ? 1 : 0) : 0, used ? e.length : 0, info(kv.first.rank).srcLength});
            delete kv.second;
        }
        M._entries.clear();
        M._reference = nullptr;
        indexEntries();
    }
    ~RunMachine() { // the entries go back to the block the other paths (and the next sequence) go on with
        if (!owner)
            return;
        for (size_t i = 0; i < erank.size(); ++i) {
            const RankInfo &ri = (*rankInfo)[(size_t)erank[i]];
            Entry *e = new Entry;
            e->genome = ri.genome;
            e->nameId = (uint32_t)ri.nameId;
            e->name = M._names[e->nameId];
            e->srcLength = ri.srcLength;
            if (elast[i] == block) {
                e->length = ent[i].len;
                e->start = (ent[i].next >> 1) - ent[i].len;
                e->strand = (ent[i].next & 1) ? '-' : '+';
                e->lastUsed = 0; // (whatever it was: the next block's resetEntries sets it to 0)
            } else {
                e->start = NULL_INDEX;
                e->length = 0;
                e->strand = '+';
                e->lastUsed = (short)(block - 1 - elast[i]);
            }
            M._entries.insert(M._entries.end(), Entries::value_type(Key{erank[i], ri.genome, ri.seq}, e));
        }
        M._reference = nullptr; // (the next block begins with resetEntries)
        M._refIndex = NULL_INDEX;
        if (batch)
            giveBatch(std::move(batch));
    }
    const RankInfo &info(int32_t rank) {
        RankInfo &ri = (*rankInfo)[(size_t)rank];
        if (ri.nameId < 0) { // initEntry's name (halMafBlock.cpp:84-112)
            const int genome = M._rankGenome[(size_t)rank], seq = M._rankSeq[(size_t)rank];
            const GenomeTables &G = img.genomes[(size_t)genome];
            const SeqInfo &S = G.seqs[(size_t)seq];
            const std::string name = M._ucscNames ? G.name + "." + S.name : S.name;
            auto it = M._nameIds.find(name);
            if (it == M._nameIds.end()) {
                it = M._nameIds.emplace(name, (uint32_t)M._names.size()).first;
                M._names.push_back(name);
            }
            ri.nameId = it->second;
            ri.rowHead = "s\t" + name + "\t";
            ri.rowTail = "\t" + std::to_string(S.length) + "\t";
            ri.srcLength = S.length;
            ri.seqStart = S.start;
            ri.genome = genome;
            ri.seq = seq;
        }
        return ri;
    }
    // device rows -> PRows, every column's sorted by rank (stable): done by the thread that fetches, beside the walk
    static void describe(const Image &img, const std::vector<std::vector<int>> &rank, PRow &p, int genome, int64_t pos, bool rev, uint32_t ord) {
        const GenomeTables &G = img.genomes[(size_t)genome];
        const int s = G.seqs.size() == 1 ? 0 : G.seqIndexBySite(pos);
        const SeqInfo &S = G.seqs[(size_t)s];
        const int64_t at = pos - S.start;
        p.key = rev ? ((S.length - 1 - at) << 1) | 1 : at << 1; // initEntry with a base (halMafBlock.cpp:84-112): start and strand
        p.rank = rank[(size_t)genome][(size_t)s];
        p.ord = ord;
    }
    // by sequence, a sequence's bases in the walk's order (a handful of rows: insertion sort)
    static void sortColumn(PRow *r, size_t n) {
        for (size_t i = 1; i < n; ++i) {
            if (r[i - 1].rank < r[i].rank || (r[i - 1].rank == r[i].rank && r[i - 1].ord <= r[i].ord))
                continue;
            const PRow x = r[i];
            size_t j = i;
            for (; j > 0 && (r[j - 1].rank > x.rank || (r[j - 1].rank == x.rank && r[j - 1].ord > x.ord)); --j)
                r[j] = r[j - 1];
            r[j] = x;
        }
    }
    void addKeys(const PRow *rows, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            if ((i > 0 && rows[i].rank == rows[i - 1].rank) || inKeys[(size_t)rows[i].rank])
                continue;
            inKeys[(size_t)rows[i].rank] = 1;
            size_t ki = 0;
            while (ki < keys.size() && keys[ki] < rows[i].rank)
                ++ki;
            keys.insert(keys.begin() + (std::ptrdiff_t)ki, rows[i].rank);
        }
    }
    void defragment(const PRow *rows, size_t n) { // ColumnIterator::defragment (halColumnIterator.cpp:193-208): keys without bases go
        for (const int32_t k : keys)
            inKeys[(size_t)k] = 0;
        keys.clear();
        for (size_t i = 0; i < n; ++i)
            if (i == 0 || rows[i].rank != rows[i - 1].rank) {
                keys.push_back(rows[i].rank);
                inKeys[(size_t)rows[i].rank] = 1;
            }
    }
    void insertEntry(size_t at, int32_t rank, bool forKey = false) { // an empty entry for the sequence (initEntry without a base)
        erank.insert(erank.begin() + (std::ptrdiff_t)at, rank);
        ekey.insert(ekey.begin() + (std::ptrdiff_t)at, forKey ? 1 : 0);
        elast.insert(elast.begin() + (std::ptrdiff_t)at, block - 1);
        ent.insert(ent.begin() + (std::ptrdiff_t)at, Ent{0, 0, info(rank).srcLength});
        indexEntries();
    }
    void indexEntries() { // firstOf[rank]: the first entry of the sequence (of a sequence without entries: anything)
        const size_t ne = erank.size();
        for (size_t i = 0; i < ne; ++i)
            if (i == 0 || erank[i] != erank[i - 1])
                firstOf[(size_t)erank[i]] = (int32_t)i;
    }
    uint32_t *idxRoom(size_t n) { // where the pairing of the next n bases is logged
        Batch &b = *batch;
        if (b.numIdx + n > b.rowEnt.size())
            b.rowEnt.resize(std::max(b.rowEnt.size() * 2, b.numIdx + n + 4096));
        return b.rowEnt.data() + b.numIdx;
    }
    // MafBlock::initBlock (halMafBlock.cpp:294-367) after resetEntries (:36-82); idx[i]: the entry base i is given to; room: the
    // columns the bases can go on inside their sequences
    void initBlock(const PRow *rows, size_t n, int64_t refPos, uint32_t *idx, int64_t &room) {
        MAF_TICK(1);
        ++block;
        size_t ne = erank.size();
        if (block == INT32_MAX)
            throw std::runtime_error("hal2maf: more than 2^31 blocks in one export");
        int32_t oldest = block;
        for (size_t i = 0; i < ne; ++i)
            oldest = std::min(oldest, elast[i]);
        bool changed = false;
        if (block - oldest >= 13) { // resetEntries: an entry unused for more than 10 blocks in a row is dropped
            size_t w = 0;
            for (size_t i = 0; i < ne; ++i) {
                if (block - elast[i] >= 13)
                    continue;
                erank[w] = erank[i];
                ekey[w] = ekey[i];
                elast[w] = elast[i];
                ent[w] = ent[i];
                ++w;
            }
            erank.resize(w);
            ekey.resize(w);
            elast.resize(w);
            ent.resize(w);
            changed = true;
        }
        if (changed)
            indexEntries();
        // every key of the column map — the sequences of this column's bases are among them — has an entry at least (the ones
        // made here come first: a later one would move the entries the bases have been given)
        for (const int32_t k : keys) {
            const size_t at = (uint32_t)firstOf[(size_t)k];
            if (at >= erank.size() || erank[at] != k) {
                insertEntry((size_t)(std::lower_bound(erank.begin(), erank.end(), k) - erank.begin()), k, true);
                changed = true;
            }
        }
        // the i-th base of a sequence goes with the sequence's i-th entry; entries that are missing are made behind the ones it has
        size_t ei = 0;
        int32_t rank = -1;
        for (size_t i = 0; i < n; ++i) {
            if (rows[i].rank != rank) {
                rank = rows[i].rank;
                ei = (uint32_t)firstOf[(size_t)rank];
            } else if (++ei == erank.size() || erank[ei] != rank) {
                insertEntry(ei, rank);
                changed = true;
            }
            idx[i] = (uint32_t)ei;
            room = std::min(room, ent[ei].src - (rows[i].key >> 1));
        }
        ne = erank.size();
#ifdef HGX_HOST_PROFILE
        if (!sliced) { // (plain words: the walks over slices run side by side)
            g_mafTicks[8] += ne;
            g_mafTicks[9] += n;
            g_mafTicks[10] += keys.size();
        }
#endif
        // the entries' ranks: logged once per change of the set (most blocks have the entries of the block before)
        if (changed || !entsLogged) {
            lastFirstEnt = (uint32_t)batch->entRank.size();
            batch->entRank.insert(batch->entRank.end(), erank.begin(), erank.end());
            entsLogged = true;
        }
        cur.firstEnt = lastFirstEnt;
        cur.numEnts = (uint32_t)ne;
        cur.firstEvent = (uint32_t)batch->events.size();
        cur.numEvents = 0;
        size_t r = refHint; // (where the reference's entry was in the block before)
        if (r >= ne || erank[r] != refRank) {
            r = 0;
            while (r < ne && erank[r] < refRank)
                ++r;
            if (r == ne || erank[r] != refRank)
                r = 0;
        } else {
            while (r > 0 && erank[r - 1] == refRank) // (several entries of the sequence: the first one)
                --r;
        }
        refHint = r;
        cur.refEnt = ne == 0 ? -1 : (int32_t)r;
        cur.refIndex = ne != 0 && erank[r] == refRank ? refPos : NULL_INDEX;
    }
    // MafBlock::canAppendColumn (halMafBlock.cpp:401-450) with appendColumn's pairing (:370-395): the i-th base of a sequence
    // goes with the i-th entry of the sequence; an entry that has bases goes on only where it ended, on its strand, below the length limit
    bool pair(const PRow *rows, size_t n, uint32_t *idx, int64_t &room) {
        MAF_TICK(2);
        const size_t ne = erank.size();
        const int32_t *er = erank.data();
        const Ent *en = ent.data();
        const int32_t *el = elast.data(), blk = block;
        const int64_t maxLength = M._maxBlockLength;
        int64_t most = room;
        size_t ei = 0;
        int32_t rank = -1;
        // (a walk over a slice notes where the answer hangs on an entry made for a key of the column map — RunMachine::KeyUse — and
        // only there: a sequence without an entry does not decide a column that cannot be appended anyway, nor does a key's entry)
        const bool noting = sliced && logging;
        const size_t notedFrom = keyUses.size();
        int32_t missing = -1; // a sequence without an entry that a key's entry would serve: the answer is no; the rest is looked at to see whether that alone decides
        for (size_t i = 0; i < n; ++i) {
            if (rows[i].rank != rank) {
                rank = rows[i].rank;
                ei = (uint32_t)firstOf[(size_t)rank];
            } else {
                if (rank == missing) { // (a second base of the sequence: one entry of a key's would not do either)
                    keyUses.resize(notedFrom);
                    return false;
                }
                ++ei;
            }
            if (ei >= ne || er[ei] != rank) {
                if (noting && missing < 0 && (i == 0 || rows[i - 1].rank != rank) && mayBeKey(rank)) {
                    missing = rank;
                    continue;
                }
                keyUses.resize(notedFrom);
                return false;
            }
            if (noting && ekey[ei] && (i == 0 || rows[i - 1].rank != rank))
                keyUses.push_back(KeyUse{lastSeen[(size_t)rank], (int64_t)numBlocks, true}); // (kept if the column is appended)
            const Ent &e = en[ei];
            const int64_t key = rows[i].key;
            if (el[ei] == blk) {
                if (e.len >= maxLength || e.next != key) {
                    keyUses.resize(notedFrom);
                    return false;
                }
                most = std::min(most, maxLength - e.len); // (appendColumn up to the length limit)
            }
            most = std::min(most, e.src - (key >> 1)); // (the columns the base can go on inside its sequence, itself among them)
            idx[i] = (uint32_t)ei;
        }
        if (missing >= 0) { // nothing else stands against the column: with an entry for this sequence it would have been appended
            keyUses.push_back(KeyUse{lastSeen[(size_t)missing], (int64_t)numBlocks, false});
            return false;
        }
        room = most;
        return true;
    }
    // could the sequence be a key of the column map for SOME count of blocks (it was seen within the last thousand blocks, or this walk
    // does not reach back that far)?
    bool mayBeKey(int32_t rank) const {
        const int64_t seen = lastSeen[(size_t)rank], at = (int64_t)numBlocks;
        return seen == INT64_MIN ? at - firstBlock <= 1000 : at - seen <= 1000;
    }
    void endBlock() {
        cur.numEvents = (uint32_t)batch->events.size() - cur.firstEvent;
        batch->blocks.push_back(cur);
    }
    // the column with these bases at reference position refPos and up to left - 1 columns behind it that continue it base by base,
    // through MafExport::convertSequence's loop body (halMafExport.cpp:60-79); returns how many columns were placed: as many
    // as fit before a block-length limit (canAppendColumn: length >= maxLength breaks) or the end of a row's sequence
    int64_t place(const PRow *rows, size_t n, int64_t left, int64_t refPos) {
        uint32_t *idx = idxRoom(n);
        // (an entry without bases takes one column at least; a column without bases limits nothing)
        int64_t k = n == 0 ? left : std::min(left, std::max<int64_t>(1, M._maxBlockLength));
        if (appendCount == 0) {
            initBlock(rows, n, refPos, idx, k);
        } else if (!pair(rows, n, idx, k)) {
            endBlock();
            if (numBlocks++ % 1000 == 0)
                defragment(rows, n);
            if (sliced) {
                if (!logging && refPos >= logFrom && refPos >= stopAt) {
                    // (no block begins inside the slice: the block that began in front of it ends behind it)
                    emptySlice = true;
                    emptyAt = refPos;
                    stopped = true;
                    return 0;
                }
                if (!logging && refPos >= logFrom && expectCount >= 0 && (int64_t)numBlocks != expectCount) {
                    // (the run-up held another number of blocks than it was reckoned with: the count it was told at its beginning is put
                    // right by the difference and the run-up walked again — walkSliced)
                    miscounted = true;
                    stopped = true;
                    return 0;
                }
                if (!logging && refPos >= logFrom) { // the slice's first block begins: the run-up's log goes
                    std::vector<std::unique_ptr<PRow[]>> keep;
                    if (!batch->extra.empty() && batch->extra.back().get() == rows)
                        keep.push_back(std::move(batch->extra.back())); // (made by advance for this very column)
                    batch->reset();
                    batch->extra = std::move(keep);
                    entsLogged = false;
                    logging = true;
                    startPending = true;
                    idx = idxRoom(n);
                } else if (logging && refPos >= stopAt && !stopped) { // the next slice's first block begins: nothing of it is kept
                    stopMark = LogMark{batch->blocks.size(), batch->events.size(), batch->numIdx, batch->entRank.size(), batch->extra.size()};
                    stopped = true;
                } else if (!logging) {
                    ++warmBlocks;
                }
            } else if (batch->blocks.size() >= 32768) {
                flush(rows);
                idx = idxRoom(n);
            }
            initBlock(rows, n, refPos, idx, k);
        }
        MAF_TICK(3);
        Ent *en = ent.data();
        int32_t *el = elast.data();
        const int32_t blk = block;
        for (size_t i = 0; i < n; ++i) { // appendColumn, k columns at once
            Ent &e = en[idx[i]];
            if (el[idx[i]] != blk) {
                el[idx[i]] = blk;
                e.len = 0;
                e.next = rows[i].key;
                ekey[idx[i]] = 0;
                lastSeen[(size_t)rows[i].rank] = (int64_t)numBlocks; // (the block being made, counted over the export)
            }
            e.len += k;
            e.next += 2 * k;
        }
        batch->events.push_back(EventLog{k, rows, (uint32_t)batch->numIdx, (uint32_t)n});
        batch->numIdx += n;
        appendCount += (size_t)k;
        if (startPending) {
            startPending = false;
            startSnap = snapshot(refPos);
            numBlocksAtStart = numBlocks;
        }
        if (stopped && endSnap.column < 0) {
            endSnap = snapshot(refPos);
            // (what the block that begins here has logged goes: it is the next slice's)
            batch->events.resize(stopMark.events);
            batch->numIdx = stopMark.numIdx;
            batch->entRank.resize(stopMark.entRank);
            if (batch->extra.size() > stopMark.extra)
                batch->extra.resize(stopMark.extra);
        }
        return k;
    }
    // The walk over chunk c from its head number hk0 on: MafExport::convertSequence's loop (halMafExport.cpp:46-81) over the
    // run-compressed columns — a head's rows come from the device, the columns up to the next head continue it base by base.
    // Ends with the chunk, or (a slice's walk) where the next slice's first block has begun.
    void walkChunk(const std::shared_ptr<Chunk> &c, size_t hk0, int64_t startPosition) {
        chunk = c;
        if (!sliced)
            batch->chunks.push_back(c);
        const int64_t n = c->n;
        size_t hk = hk0;
        const size_t numChunkHeads = c->headOff.size() - 1;
        for (int64_t i = hk0 == 0 ? 0 : (int64_t)c->headCol[hk0]; i < n && !stopped;) {
            if (c->head[(size_t)i] == 2) { // --unique: a column the iterator does not walk (nextFreeIndex passes over it)
                ++i;
                continue;
            }
            // head column i: its rows come from the device; the columns up to the next head continue it
            if (sliced && logging && !stopped && startPosition + c->done + i >= stopAt) {
                preStop = saveState(); // (the next slice's first block begins at this head or one of the next few)
                preStopChunk = curChunkIndex;
                preStopHead = hk;
                preStopValid = true;
            }
            const PRow *rows = c->rows.get() + c->headOff[hk];
            const size_t nr = c->headOff[hk + 1] - c->headOff[hk];
            ++hk;
            if (hk + 6 < numChunkHeads) { // (the rows were written by other cores a moment ago: asked for a few heads ahead of their use)
                const char *ahead = (const char *)(c->rows.get() + c->headOff[hk + 5]);
                __builtin_prefetch(ahead);
                __builtin_prefetch(ahead + 64);
                __builtin_prefetch(ahead + 128);
                __builtin_prefetch(ahead + 192);
            }
            if (c->head[(size_t)i] == 3) { // --unique: walked, not written (a reference base left of the range): its sequences stay
                addKeys(rows, nr);         // behind as keys of the column map (halColumnIterator.cpp:822-826)
                for (size_t r = 0; r < nr; ++r)
                    lastSeen[(size_t)rows[r].rank] = (int64_t)numBlocks;
                irregularKeys = true;
                ++i;
                continue;
            }
            int64_t left = 1; // (the columns up to the next head, eight bytes of the marks at a time)
            {
                const uint8_t *hp = c->head.data();
                int64_t at = i + 1;
                for (; at + 8 <= n; at += 8) {
                    uint64_t w;
                    memcpy(&w, hp + at, 8);
                    if (w) {
                        at += __builtin_ctzll(w) >> 3;
                        break;
                    }
                }
                while (at < n && !hp[at])
                    ++at;
                left = at - i;
            }
            int64_t col = i;
            for (;;) {
                addKeys(rows, nr);
                const int64_t k = place(rows, nr, left, startPosition + c->done + col);
                left -= k;
                col += k;
                if (left == 0 || stopped)
                    break;
                rows = advance(rows, nr, k); // a block-length break or a sequence end: an ordinary column next
            }
            i = col;
        }
    }
    // the bases k columns on (a column that has to go through the per-column logic in the middle of a run)
    const PRow *advance(const PRow *rows, size_t n, int64_t k) {
        std::unique_ptr<PRow[]> next(new PRow[n]);
        for (size_t i = 0; i < n; ++i) { // (k bases on along its strand: that may be the genome's next sequence)
            const RankInfo &ri = info(rows[i].rank);
            const bool rev = (rows[i].key & 1) != 0;
            const int64_t q = rows[i].key >> 1, pos = ri.seqStart + (rev ? ri.srcLength - 1 - q : q);
            describe(img, M._rank, next[i], ri.genome, pos + (rev ? -k : k), rev, rows[i].ord);
        }
        sortColumn(next.get(), n);
        batch->extra.push_back(std::move(next));
        return batch->extra.back().get();
    }
    // hands the batch to the rendering threads (beside the walk: the previous batch is waited for first); current: the column
    // the next block begins with
    void flush(const PRow *current = nullptr);
    static bool renderOnDevice(const Batch &work, const std::vector<RankInfo> &ranks, const hgx_alignment *al, bool keepEmptyRefBlocks, char *&text,
                               size_t &bytes);
};

// a row's start and length as decimal text, two digits at a time (std::to_chars was a sixth of the rendering threads' time)
static inline char *mafNumber(char *o, int64_t v) {
    static const struct Pairs {
        char d[200];
        Pairs() {
            for (int i = 0; i < 100; ++i) {
                d[2 * i] = (char)('0' + i / 10);
                d[2 * i + 1] = (char)('0' + i % 10);
            }
        }
    } pairs;
    uint64_t u = (uint64_t)v;
    if (v < 0) {
        *o++ = '-';
        u = 0 - u;
    }
    char tmp[24];
    int n = 0;
    while (u >= 100) {
        const uint64_t q = u / 100;
        const unsigned r = (unsigned)(u - q * 100);
        tmp[n++] = pairs.d[2 * r + 1];
        tmp[n++] = pairs.d[2 * r];
        u = q;
    }
    if (u >= 10) {
        tmp[n++] = pairs.d[2 * u + 1];
        tmp[n++] = pairs.d[2 * u];
    } else {
        tmp[n++] = (char)('0' + u);
    }
    while (n > 0)
        *o++ = tmp[--n];
    return o;
}

namespace {
struct TextBuffer {
    char *data = nullptr;
    size_t len = 0, cap = 0;
    TextBuffer() = default;
    TextBuffer(const TextBuffer &) = delete;
    TextBuffer &operator=(const TextBuffer &) = delete;
    TextBuffer(TextBuffer &&o) noexcept : data(o.data), len(o.len), cap(o.cap) {
        o.data = nullptr;
        o.len = o.cap = 0;
    }
    ~TextBuffer() { free(data); }
    char *room(size_t n) {
        if (len + n > cap) {
            cap = std::max(cap * 2, len + n + (1u << 20));
            data = (char *)realloc(data, cap);
            if (!data)
                throw std::bad_alloc();
        }
        return data + len;
    }
};
} // namespace

static int renderThreads() { // hal2maf's rendering threads per batch (HGX_MAF_RENDER_THREADS; the walk is one thread beside them)
    const char *e = getenv("HGX_MAF_RENDER_THREADS"); // (read per batch: a process can try several settings)
    // (48: with the walk at a half of its old time the rendering is what it waits for — config 3 in one process on the GPU box:
    // 24 threads 0.67 s, 32 0.74-0.75, 48 0.63, 64 0.61, profiles/r04y_gpu_maf_diag_threads.txt; more threads also disturb the walk more)
    return e ? std::max(1, atoi(e)) : 48;
}
namespace {
struct LastExport { // the last run-compressed export of this process: who walked it and how long the parts took
    std::mutex mu;
    std::string json = "null";
};
LastExport &lastExport() {
    static LastExport *l = new LastExport;
    return *l;
}
} // namespace
std::string mafLastExportInfo() {
    std::lock_guard<std::mutex> lock(lastExport().mu);
    return lastExport().json;
}
static bool deviceRenderWanted();
static int rendersInFlight() { // batches rendered at a time (HGX_MAF_RENDERS_IN_FLIGHT): their threads share the host's
    const char *e = getenv("HGX_MAF_RENDERS_IN_FLIGHT");
    if (e)
        return std::max(1, atoi(e));
    // (four where the host has the threads — and wherever the text is rendered on the device: a batch's task then waits for
    // the device most of its time, and config 3 on the GPU box's 16 CPUs went from 0.30-0.35 s to 0.26-0.30 s with four)
    return hostThreads() >= 64 || deviceRenderWanted() ? 4 : 2;
}
static size_t describeThreads(size_t heads) { // the threads that describe and sort a device batch's rows (HGX_MAF_DESCRIBE_THREADS)
    if (const char *e = getenv("HGX_MAF_DESCRIBE_THREADS"))
        return (size_t)std::max(1, atoi(e));
    return heads >= 32768 ? 8 : heads >= 4096 ? 4 : 1;
}
// (HGX_MAF_TIMING: the CPU time the export's stages used, summed over their threads — what a host with a CPU quota has to pay for)
static std::atomic<long long> g_cpuWalkNs{0}, g_cpuRenderNs{0}, g_cpuCopyNs{0}, g_cpuDescribeNs{0}, g_cpuDeviceNs{0};
struct CpuScope {
    std::atomic<long long> &to;
    long long t0;
    static long long now() {
        timespec ts;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
    }
    explicit CpuScope(std::atomic<long long> &a) : to(a), t0(now()) {}
    ~CpuScope() { to.fetch_add(now() - t0, std::memory_order_relaxed); }
};
static double g_mafFlush = 0; // (HGX_MAF_TIMING: the walk's thread in flush(): names, hand-over, without the wait for the batch before)
static double g_mafRenderWait = 0; // (HGX_MAF_TIMING: how long the walk stood waiting for the batch before to be rendered)
static bool deviceRenderWanted() { // HGX_MAF_DEVICE_RENDER=0: the rendering threads whatever the handle
    const char *e = getenv("HGX_MAF_DEVICE_RENDER");
    return !(e && e[0] == '0');
}
static size_t deviceRenderMinBlocks() { // (a handful of blocks is not worth the copies; the tests send every batch: HGX_MAF_DEVICE_RENDER_MIN_BLOCKS=1)
    const char *e = getenv("HGX_MAF_DEVICE_RENDER_MIN_BLOCKS");
    return e ? (size_t)std::max(1, atoi(e)) : 64;
}
// A batch's log in the layouts of hgx_maf_render_kernels.hpp, in one page-locked block, and the call (hgx_columns.hip:
// mafRenderDevice).  false: the device did not take it (the rendering threads do).
bool MafExport::RunMachine::renderOnDevice(const Batch &work, const std::vector<RankInfo> &ranks, const hgx_alignment *al, bool keepEmptyRefBlocks,
                                           char *&text, size_t &bytes) {
    static_assert(sizeof(PRow) == sizeof(MafRenderRow), "PRow is what the device reads");
    const size_t nb = work.blocks.size(), ne = work.events.size();
    size_t numRows = 0, slots = 0;
    for (const EventLog &e : work.events)
        numRows += e.nRows;
    for (const BlockLog &b : work.blocks)
        slots += b.numEnts;
    if (numRows >= (1ull << 32) || slots >= (1ull << 31) || work.numIdx >= (1ull << 32))
        return false;
    // the sequences the batch's entries belong to (only those: the walk goes on looking others up while this batch is rendered)
    std::vector<MafRenderRank> table(ranks.size());
    memset(table.data(), 0, table.size() * sizeof(MafRenderRank));
    std::string chars;
    for (int32_t r : work.entRank) {
        MafRenderRank &T = table[(size_t)r];
        if (T.headLen)
            continue;
        const RankInfo &ri = ranks[(size_t)r];
        T.seqStart = ri.seqStart;
        T.srcLength = ri.srcLength;
        T.genome = ri.genome;
        T.headOff = (uint32_t)chars.size();
        T.headLen = (uint32_t)ri.rowHead.size();
        chars += ri.rowHead;
        T.tailOff = (uint32_t)chars.size();
        T.tailLen = (uint32_t)ri.rowTail.size();
        chars += ri.rowTail;
    }
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t oBlocks = 0, oEvents = oBlocks + pad(nb * sizeof(MafRenderBlock)), oRows = oEvents + pad(ne * sizeof(MafRenderEvent)),
                 total = oRows + pad(numRows * sizeof(MafRenderRow));
    struct Block {
        char *p;
        ~Block() { hostBlockGive(p); }
    } stage{static_cast<char *>(hostBlockTake(std::max<size_t>(total, 64)))};
    MafRenderBlock *blocks = reinterpret_cast<MafRenderBlock *>(stage.p + oBlocks);
    MafRenderEvent *events = reinterpret_cast<MafRenderEvent *>(stage.p + oEvents);
    MafRenderRow *rows = reinterpret_cast<MafRenderRow *>(stage.p + oRows);
    uint32_t slot = 0;
    for (size_t b = 0; b < nb; ++b) {
        const BlockLog &B = work.blocks[b];
        blocks[b] = MafRenderBlock{B.firstEnt, B.numEnts, B.firstEvent, B.numEvents, B.refEnt, slot, B.refIndex};
        slot += B.numEnts;
    }
    uint32_t at = 0;
    for (size_t e = 0; e < ne; ++e) {
        const EventLog &E = work.events[e];
        events[e] = MafRenderEvent{E.k, at, E.firstIdx, E.nRows, 0};
        memcpy(rows + at, E.rows, (size_t)E.nRows * sizeof(PRow));
        at += E.nRows;
    }
    MafRenderInput in;
    in.blocks = blocks;
    in.numBlocks = nb;
    in.entRank = work.entRank.data();
    in.numEntRank = work.entRank.size();
    in.events = events;
    in.numEvents = ne;
    in.rowEnt = work.rowEnt.data();
    in.numRowEnt = work.numIdx;
    in.rows = rows;
    in.numRows = numRows;
    in.ranks = table.data();
    in.numRanks = table.size();
    in.chars = chars.data();
    in.numChars = chars.size();
    in.slots = slots;
    in.keepEmptyRefBlocks = keepEmptyRefBlocks;
    return mafRenderDevice(const_cast<hgx_alignment *>(al), in, text, bytes);
}

void MafExport::RunMachine::flush(const PRow *current) {
    {
        // (up to rendersInFlight() batches are being rendered at a time; a batch whose threads are done waits for its turn to write)
        const auto tw = std::chrono::steady_clock::now();
        if (M._pendingWrite.valid())
            M._pendingWrite.get();
        const size_t inFlight = batch->blocks.empty() ? 0 : (size_t)rendersInFlight();
        while (M._pendingWrites.size() > (inFlight ? inFlight - 1 : 0)) {
            std::future<void> f = std::move(M._pendingWrites.front()); // (out of the deque first: get() may throw, and an invalid future must not stay behind)
            M._pendingWrites.pop_front();
            f.get();
        }
        g_mafRenderWait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
    }
    if (batch->blocks.empty())
        return;
    const auto tFlush = std::chrono::steady_clock::now();
    struct FlushTime {
        std::chrono::steady_clock::time_point t0;
        ~FlushTime() { g_mafFlush += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } flushTime{tFlush};
    std::shared_ptr<Batch> work(batch.release(), [](Batch *b) { giveBatch(std::unique_ptr<Batch>(b)); }); // (back to the pool when rendered)
    batch = takeBatch();
    entsLogged = false;
    if (chunk)
        batch->chunks.push_back(chunk); // (still being walked)
    if (!work->extra.empty() && work->extra.back().get() == current) { // (made by advance for the block to come: no event of this batch points to it)
        batch->extra.push_back(std::move(work->extra.back()));
        work->extra.pop_back();
    }
    // (the names by address, taken here: the walk may add names while the batch is rendered; a deque's elements stay put)
    auto names = std::make_shared<std::vector<const std::string *>>();
    names->reserve(M._names.size());
    for (const std::string &n : M._names)
        names->push_back(&n);
    std::shared_ptr<std::vector<RankInfo>> ranks = rankInfo;
    const hgx_alignment *al = M._al;
    std::ostream *out = &os;
    const bool keepEmptyRefBlocks = M._keepEmptyRefBlocks;
#ifdef HGX_HOST_PROFILE
    if (getenv("HGX_MAF_NO_RENDER"))
        return;
#endif
    const size_t ticket = M._ticketsIssued++;
    std::shared_ptr<WriteOrder> order = M._writeOrder;
    M._pendingWrites.push_back(std::async(std::launch::async, [work, names, ranks, al, out, keepEmptyRefBlocks, ticket, order]() {
        struct Turn { // (whatever happens to this batch, the batches behind it get their turn)
            WriteOrder &o;
            size_t t;
            ~Turn() { o.done(t); }
        } turn{*order, ticket};
        static const char fwd[17] = "acgtn\0\0\0ACGTN\0\0\0";
        static const char rc[17] = "tgcan\0\0\0TGCAN\0\0\0";
        static const PairTable fwd2(fwd, false), rc2(rc, true);
        const size_t nb = work->blocks.size();
        unsigned nt = hostThreads();
        nt = std::max(1u, std::min(nt ? nt : 1u, (unsigned)renderThreads()));
        nt = std::max(1u, std::min(nt, (hostThreads() + (unsigned)rendersInFlight() - 1) / (unsigned)rendersInFlight()));
        if (nb < 256)
            nt = 1;
        BulkSink *const sink = dynamic_cast<BulkSink *>(out->rdbuf());
        // The text on the device where there is one (hgx_maf_render_kernels.hpp): the log goes over as it is, the text comes back
        if (nb >= deviceRenderMinBlocks() && al->dev && deviceRenderWanted()) {
            char *text = nullptr;
            size_t bytes = 0;
            bool made = false;
            {
                CpuScope cpu(g_cpuRenderNs);
                made = renderOnDevice(*work, *ranks, al, keepEmptyRefBlocks, text, bytes);
            }
            if (made) {
                struct Give {
                    char *p;
                    ~Give() { hostBlockGive(p); }
                } give{text};
                order->wait(ticket); // (the batches' texts in the batches' order)
                char *dst = sink && bytes >= (1u << 20) ? sink->room(bytes) : nullptr;
                if (!dst) {
                    out->write(text, (std::streamsize)bytes);
                } else {
                    // (copied into place by a few threads: one thread's copy of a batch's thirty megabytes into fresh pages is ten
                    // milliseconds the next batch's text waits for)
                    const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(nt, 8u), bytes >> 22));
                    auto copy = [&](unsigned t) {
                        CpuScope cpu(g_cpuCopyNs);
                        const size_t a = bytes * t / parts, b = bytes * (t + 1) / parts;
                        memcpy(dst + a, text + a, b - a);
                    };
                    std::vector<std::thread> th;
                    for (unsigned t = 1; t < parts; ++t)
                        th.emplace_back(copy, t);
                    copy(0);
                    for (std::thread &x : th)
                        x.join();
                }
                return;
            }
        }
        // (the rendering threads' buffers are kept from batch to batch: thirty megabytes of fresh pages per batch were as many page
        // faults again as the text's own)
        struct BufferPool {
            std::mutex mu;
            std::vector<std::unique_ptr<std::vector<TextBuffer>>> idle;
        };
        static BufferPool *pool = new BufferPool;
        std::unique_ptr<std::vector<TextBuffer>> held;
        {
            std::lock_guard<std::mutex> lock(pool->mu);
            if (!pool->idle.empty()) {
                held = std::move(pool->idle.back());
                pool->idle.pop_back();
            }
        }
        if (!held)
            held.reset(new std::vector<TextBuffer>());
        if (held->size() < nt)
            held->resize(nt);
        std::vector<TextBuffer> &text = *held;
        for (TextBuffer &t : text)
            t.len = 0;
        auto render = [&](unsigned t) {
            TextBuffer &buf = text[t];
            struct RowOut {
                int64_t start, length;
                uint32_t firstSeg, numSegs;
                bool rev;
            };
            std::vector<RowOut> rows;
            std::vector<Entry::Seg> segs;
            std::vector<const PRow *> given;
            // A block lists every entry the walk still remembered (MafBlock keeps an entry for thirteen blocks after its last base:
            // 29 entries a block on config 3) and gives bases to six or seven of them: only those are rows of the text.  stamp[j] == b + 1:
            // entry j was given a base in block b and is number slot[j] among them.
            std::vector<uint32_t> stamp, slot, touched;
            const std::vector<uint8_t> *dnaOf = nullptr;
            for (size_t b = nb * t / nt; b < nb * (t + 1) / nt; ++b) {
                const BlockLog &B = work->blocks[b];
                const EventLog *ev = work->events.data() + B.firstEvent;
                rows.clear();
                segs.clear();
                touched.clear();
                int64_t columns = 0;
                for (uint32_t e = 0; e < B.numEvents; ++e)
                    columns += ev[e].k;
                if (stamp.size() < B.numEnts) {
                    stamp.resize(B.numEnts, 0);
                    slot.resize(B.numEnts, 0);
                }
                const uint32_t mark = (uint32_t)(b - nb * t / nt) + 1;
                for (uint32_t e = 0; e < B.numEvents; ++e) {
                    const uint32_t *idx = work->rowEnt.data() + ev[e].firstIdx;
                    for (uint32_t r = 0; r < ev[e].nRows; ++r)
                        if (stamp[idx[r]] != mark) {
                            stamp[idx[r]] = mark;
                            touched.push_back(idx[r]);
                        }
                }
                // (the text's order is the entries' order)
                for (size_t i = 1; i < touched.size(); ++i) {
                    const uint32_t x = touched[i];
                    size_t j = i;
                    for (; j > 0 && touched[j - 1] > x; --j)
                        touched[j] = touched[j - 1];
                    touched[j] = x;
                }
                const size_t nT = touched.size();
                for (size_t i = 0; i < nT; ++i)
                    slot[touched[i]] = (uint32_t)i;
                // which base every such entry was given at every event (the walk logs the other direction)
                given.assign((size_t)B.numEvents * nT, nullptr);
                for (uint32_t e = 0; e < B.numEvents; ++e) {
                    const uint32_t *idx = work->rowEnt.data() + ev[e].firstIdx;
                    for (uint32_t r = 0; r < ev[e].nRows; ++r)
                        given[(size_t)e * nT + slot[idx[r]]] = ev[e].rows + r;
                }
                // appendColumn / updateEntry (halMafBlock.cpp:114-138, 370-395) for every entry that was given a base, the row kept as
                // runs; rows[i] belongs to entry touched[i]
                for (size_t i = 0; i < nT; ++i) {
                    RowOut r{NULL_INDEX, 0, (uint32_t)segs.size(), 0, false};
                    const RankInfo &ri = (*ranks)[(size_t)work->entRank[B.firstEnt + touched[i]]];
                    dnaOf = &al->img.genomes[(size_t)ri.genome].dna;
                    for (uint32_t e = 0; e < B.numEvents; ++e) {
                        const PRow *p = given[(size_t)e * nT + i];
                        const int64_t k = ev[e].k;
                        const uint8_t kind = !p ? 0 : ((p->key & 1) ? 2 : 1);
                        if (p) {
                            if (r.start == NULL_INDEX) {
                                r.start = p->key >> 1;
                                r.rev = (p->key & 1) != 0;
                            }
                            r.length += k;
                        }
                        // (the base's genome coordinate: the packed DNA is read there)
                        const int64_t pos = !p ? 0 : ri.seqStart + ((p->key & 1) ? ri.srcLength - 1 - (p->key >> 1) : p->key >> 1);
                        if (segs.size() > r.firstSeg) {
                            Entry::Seg &l = segs.back();
                            if (l.kind == kind && (int64_t)l.n + k < INT32_MAX &&
                                (kind == 0 || (kind == 1 ? pos == l.pos + l.n : pos == l.pos - l.n))) {
                                l.n += (int32_t)k;
                                continue;
                            }
                        }
                        // (the packed bases lie anywhere in fifty megabytes a genome: asked for now, they are there when the row is
                        // written — the rows' misses overlap instead of following one another)
                        if (kind && !dnaOf->empty())
                            __builtin_prefetch(dnaOf->data() + (pos >> 1));
                        for (int64_t left = k, at = pos; left > 0;) { // (a run longer than 2^31 - 1 columns is split)
                            const int64_t m = std::min<int64_t>(left, INT32_MAX - 1);
                            segs.push_back(Entry::Seg{at, (int32_t)m, kind});
                            at += kind == 2 ? -m : m;
                            left -= m;
                        }
                    }
                    r.numSegs = (uint32_t)segs.size() - r.firstSeg;
                    rows.push_back(r);
                }
                if (B.refEnt < 0)
                    continue;
                const uint32_t ref = (uint32_t)B.refEnt;
                const bool refGiven = stamp[ref] == mark; // (else the reference's row is all gaps)
                if (!keepEmptyRefBlocks && !refGiven)
                    continue; // referenceIsAllGaps (halMafExport.cpp:70, 85)
                // MafBlock's operator<< (halMafBlock.cpp:499-520): the reference row first, then the entries that have a start
                auto row = [&](const RowOut &r, uint32_t j, int64_t start) {
                    const RankInfo &ri = (*ranks)[(size_t)work->entRank[B.firstEnt + j]];
                    char *o = buf.room(ri.rowHead.size() + ri.rowTail.size() + 64 + (size_t)columns);
                    char *const o0 = o;
                    memcpy(o, ri.rowHead.data(), ri.rowHead.size());
                    o += ri.rowHead.size();
                    o = mafNumber(o, start);
                    *o++ = '\t';
                    o = mafNumber(o, r.length);
                    *o++ = '\t';
                    *o++ = r.rev ? '-' : '+';
                    memcpy(o, ri.rowTail.data(), ri.rowTail.size());
                    o += ri.rowTail.size();
                    const std::vector<uint8_t> &d = al->img.genomes[(size_t)ri.genome].dna;
                    const uint8_t *pk = d.data();
                    for (uint32_t g = 0; g < r.numSegs; ++g) {
                        const Entry::Seg &sg = segs[r.firstSeg + g];
                        const int64_t n = sg.n;
                        if (sg.kind == 0) {
                            memset(o, '-', (size_t)n);
                        } else if (d.empty()) {
                            memset(o, 'N', (size_t)n);
                        } else if (sg.kind == 1) { // dnaUnpack (halCommon.h:187-190), two bases per packed byte
                            int64_t p0 = sg.pos, i = 0;
                            if (i < n && (p0 & 1)) {
                                o[i++] = fwd[pk[p0 >> 1] & 0x0F];
                                ++p0;
                            }
                            for (; i + 1 < n; i += 2, p0 += 2)
                                memcpy(o + i, &fwd2.v[pk[p0 >> 1]], 2);
                            if (i < n)
                                o[i] = fwd[pk[p0 >> 1] >> 4];
                        } else { // reverse strand: walk left, complemented (reverseComplement, halCommon.h:45-75)
                            int64_t p0 = sg.pos, i = 0;
                            if (i < n && !(p0 & 1)) {
                                o[i++] = rc[pk[p0 >> 1] >> 4];
                                --p0;
                            }
                            for (; i + 1 < n; i += 2, p0 -= 2)
                                memcpy(o + i, &rc2.v[pk[p0 >> 1]], 2);
                            if (i < n)
                                o[i] = rc[pk[p0 >> 1] & 0x0F];
                        }
                        o += n;
                    }
                    *o++ = '\n';
                    buf.len += (size_t)(o - o0);
                };
                memcpy(buf.room(2), "a\n", 2);
                buf.len += 2;
                if (!refGiven) {
                    if (B.refIndex != NULL_INDEX) { // (a row of gaps, as long as the block)
                        RowOut gaps{NULL_INDEX, 0, (uint32_t)segs.size(), 0, false};
                        for (int64_t left = columns; left > 0;) {
                            const int64_t m = std::min<int64_t>(left, INT32_MAX - 1);
                            segs.push_back(Entry::Seg{0, (int32_t)m, 0});
                            left -= m;
                        }
                        gaps.numSegs = (uint32_t)segs.size() - gaps.firstSeg;
                        row(gaps, ref, B.refIndex);
                    }
                } else {
                    row(rows[slot[ref]], ref, rows[slot[ref]].start);
                }
                for (size_t i = 0; i < nT; ++i)
                    if (touched[i] != ref)
                        row(rows[i], touched[i], rows[i].start);
                *buf.room(1) = '\n';
                buf.len += 1;
            }
        };
        if (nt == 1) {
            {
                CpuScope cpu(g_cpuRenderNs);
                render(0);
            }
            order->wait(ticket);
            out->write(text[0].data, (std::streamsize)text[0].len);
        } else {
            // the threads are made once a batch: each renders its share of the blocks, waits until all have and the output has given
            // room for the batch's text at once, and copies its share there — side by side (else the shares are written one after
            // the other by this thread)
            std::mutex mu;
            std::condition_variable cv;
            unsigned rendered = 0;
            bool placed = false;
            char *dst = nullptr;
            std::vector<size_t> at(nt, 0);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([&, t]() {
                    {
                        CpuScope cpu(g_cpuRenderNs);
                        render(t);
                    }
                    std::unique_lock<std::mutex> lock(mu);
                    ++rendered;
                    cv.notify_all();
                    cv.wait(lock, [&]() { return placed; });
                    lock.unlock();
                    CpuScope cpu(g_cpuCopyNs);
                    if (dst)
                        memcpy(dst + at[t], text[t].data, text[t].len);
                });
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&]() { return rendered == nt; });
                order->wait(ticket); // (the batches' texts in the batches' order)
                size_t total = 0;
                for (unsigned t = 0; t < nt; ++t) {
                    at[t] = total;
                    total += text[t].len;
                }
                if (sink && total >= (1u << 20))
                    dst = sink->room(total);
                placed = true;
                cv.notify_all();
            }
            for (std::thread &x : th)
                x.join();
            if (!dst)
                for (unsigned t = 0; t < nt; ++t)
                    out->write(text[t].data, (std::streamsize)text[t].len);
        }
        std::lock_guard<std::mutex> lock(pool->mu);
        if (pool->idle.size() < 6)
            pool->idle.push_back(std::move(held));
    }));
}

// MafBlock's state machine over slices of the export, side by side.  Where a block begins depends on the blocks before it only
// through the entries (a memory of thirteen blocks), the column map's keys (since the last defragment) and the count of blocks
// so far (defragment runs at every thousandth) — RunMachine::Snapshot.  Every slice but the first is walked cold from some
// thousand heads in front of it, told the count of blocks in front of it as the round before found it (first round: nothing),
// and logs the blocks that begin inside it; a slice's walk goes on until the next slice's first block has begun and keeps what it
// looked like then.  When every slice began as the slice before it ended, the logs in order are the one-thread walk's, by
// induction from the first slice (which begins with the export's own state).  Counts that moved, or a run-up that did not find
// the state, cost another round for the slices behind; after a few rounds without agreement one thread walks the export.
// The export's batches as they arrive from the device stage (convertSequenceRuns' feeding thread), for the walk over slices: the walk
// begins with the first batches and takes the others in as they come, so that its rounds — and the rendering of the slices they
// settle — run beside the device stage instead of behind it.
struct MafExport::Arrivals {
    typedef RunMachine::Chunk Chunk;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::shared_ptr<Chunk>> chunks; // (under mu: the vector grows)
    std::vector<size_t> headBase{0};            // heads in front of every chunk that is there, and behind the last one
    bool complete = false;                      // no more batches
    std::exception_ptr error;                   // the feeding thread's
    double completeAt = 0;                      // seconds after `since`
    std::chrono::steady_clock::time_point since = std::chrono::steady_clock::now();
    void push(std::shared_ptr<Chunk> c) {
        std::lock_guard<std::mutex> lock(mu);
        headBase.push_back(headBase.back() + (c->headOff.size() - 1));
        chunks.push_back(std::move(c));
        cv.notify_all();
    }
    void end(std::exception_ptr e = nullptr) {
        std::lock_guard<std::mutex> lock(mu);
        complete = true;
        error = e;
        completeAt = std::chrono::duration<double>(std::chrono::steady_clock::now() - since).count();
        cv.notify_all();
    }
    // batch c, waited for; null: the export has fewer batches
    std::shared_ptr<Chunk> get(size_t c) {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&]() { return c < chunks.size() || complete; });
        if (error)
            std::rethrow_exception(error);
        return c < chunks.size() ? chunks[c] : nullptr;
    }
    // the batches that are there (and whether they are all), once there are more than `have` or they are all
    size_t waitMore(size_t have, bool &all) {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&]() { return chunks.size() > have || complete; });
        if (error)
            std::rethrow_exception(error);
        all = complete;
        return chunks.size();
    }
    std::vector<size_t> headBases() {
        std::lock_guard<std::mutex> lock(mu);
        return headBase;
    }
    void waitComplete() {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&]() { return complete; });
        if (error)
            std::rethrow_exception(error);
    }
};

bool MafExport::walkSliced(std::ostream &mafStream, Arrivals &A, int refRank, int64_t startPosition, size_t &numBlocksOut, int *roundsOut,
                           unsigned *threadsOut, double *secondsOut) {
    typedef RunMachine::Chunk Chunk;
    // slices = batches.  `there`: the batches that have arrived; S: the slices in play — a slice's walk goes on into the batch behind
    // it (until that slice's first block begins), so the last batch that is there is in play only when it is the export's last
    bool complete = false;
    size_t there = A.waitMore(1, complete);
    if (there < 2)
        return false; // (one batch: nothing to walk side by side)
    size_t S = complete ? there : there - 1;
    const auto t0 = std::chrono::steady_clock::now();
    RunMachine first(*this, mafStream, refRank); // (takes the export's entries; hands the last slice's back in the end)
    first.describeAllRanks();
    const RunMachine::MachineState initial = first.saveState();
    std::shared_ptr<std::vector<RunMachine::RankInfo>> ranks = first.rankInfo;
    // heads in front of every chunk; where a slice's run-up begins
    std::vector<size_t> headBase = A.headBases();
    const size_t RUNUP = getenv("HGX_MAF_RUNUP") ? (size_t)std::max(1, atoi(getenv("HGX_MAF_RUNUP"))) : 4096;
    // the first column of slice s (its batch is there, or the one before it is): a walk is told where the slice behind its own begins
    auto seamOf = [&](size_t s) {
        if (std::shared_ptr<Chunk> c = A.get(s))
            return startPosition + c->done;
        return (int64_t)INT64_MAX;
    };
    struct Slice {
        std::unique_ptr<RunMachine> R;
        int64_t count = 0, runup = 0;  // blocks that began in the slice / in its run-up, as its last walk found them
        int64_t toldCount = -1;        // the count its last walk was told at its run-up's beginning
        int64_t shift = 0;             // how far the count its accepted walk was told was off (nothing it decided depended on it)
        int64_t exactBase = -1;        // the count in front of it, once the slices before it are settled
        int64_t guessBase = -1;        // ... or as the last round's walks let it be told
        int64_t lastBase = -1;         // what its last cold walk was told
        bool settledEmpty = false;     // no block begins in it
        bool exactWalk = false;        // walked from the very state the slice before it stopped in
        bool exactState = false;       // its machine's count and keys are the one-thread walk's (else: right but for them)
        bool walked = false, reachedEnd = false;
        int64_t predictedCount = -1, predictedShift = 0; // (HGX_MAF_TIMING: what the count model said of this slice in the round before)
        size_t predictedLooks = 0;
    };
    std::deque<Slice> slice(S); // (grows between rounds; a round's walks each touch their own)
    // the first slice: the export's own state, its log from the first column on
    first.sliced = true;
    first.stopAt = seamOf(1);
    // (a walk that runs out of batches waits for the next one: only the export's end ends it)
    auto walkFrom = [&](RunMachine &R, size_t chunk, size_t head) {
        for (size_t c = chunk; !R.stopped; ++c) {
            std::shared_ptr<Chunk> ch = A.get(c);
            if (!ch)
                break;
            R.curChunkIndex = c;
            R.walkChunk(ch, c == chunk ? head : 0, startPosition);
        }
        if (!R.stopped && R.appendCount > 0)
            R.endBlock();
    };
    {
        CpuScope cpu(g_cpuWalkNs);
        walkFrom(first, 0, 0);
    }
    slice[0].count = (int64_t)(first.stopped ? first.endSnap.numBlocks : first.numBlocks);
    slice[0].reachedEnd = !first.stopped;
    unsigned threads = hostThreads();
    threads = std::max(1u, std::min(threads ? threads : 1u, threads >= 96 ? 64u : 32u));
    if (const char *e = getenv("HGX_MAF_WALK_THREADS"))
        threads = (unsigned)std::max(1, atoi(e));
    // the settled slices' logs go to the rendering threads in order as soon as they are settled (RunMachine::flush renders a few
    // batches at a time beside whatever this thread does next: the following round's walks)
    size_t flushedUpTo = 0, blocksFlushed = 0;
    bool flushEnded = false;
    auto flushSettled = [&](size_t upTo) {
        for (; flushedUpTo < upTo && !flushEnded; ++flushedUpTo) {
            const size_t s = flushedUpTo;
            if (s > 0 && slice[s].settledEmpty)
                continue;
            RunMachine &R = s == 0 ? first : *slice[s].R;
            flushEnded = slice[s].reachedEnd;
            blocksFlushed += R.batch->blocks.size();
            if (&R != &first) {
                first.batch = std::move(R.batch);
                R.batch.reset(new RunMachine::Batch);
            }
            first.sliced = false;
            first.flush();
            first.sliced = true;
        }
    };
    std::vector<char> todo(S, 1);
    todo[0] = 0;
    bool settled = false;
    int rounds = 0;
    size_t frontier = 1, frontierPrev = 0; // the first slice that is not settled; the last slice in front of it with blocks of its own
    slice[0].exactState = true;
    // (every round settles its frontier at least: the rounds are fewer than the slices plus the times new batches came into play)
    for (; rounds < 100000 && !settled; ++rounds) {
        std::atomic<size_t> next{1};
        std::exception_ptr failure;
        std::mutex failureMu;
        // (a walk is told the blocks in front of its slice as the round before counted them: taken before the threads start — a
        // round's walks write only their own slice's counts)
        std::vector<int64_t> baseOf(S, 0);
        for (size_t s = 1; s < S; ++s)
            baseOf[s] = slice[s].exactBase >= 0 ? slice[s].exactBase : slice[s].guessBase >= 0 ? slice[s].guessBase : baseOf[s - 1] + slice[s - 1].count;
        auto walkSlice = [&](size_t s) {
            Slice &L = slice[s];
            L.exactWalk = false;
            if (s == frontier) {
                // the first slice that is not settled: from the state the slice before it was in when it reached this one (put right
                // for the count where that slice was told another) — the one-thread walk of this slice, whatever the others guess
                const size_t q = frontierPrev;
                const RunMachine &P = q == 0 ? first : *slice[q].R;
                if (P.preStopValid) {
                    const RunMachine::MachineState &st = P.preStop;
                    L.R.reset(new RunMachine(*this, mafStream, refRank, ranks, st.numBlocks));
                    RunMachine &R = *L.R;
                    R.restoreState(st);
                    R.logFrom = seamOf(s);
                    R.stopAt = seamOf(s + 1);
                    R.logging = false;
                    walkFrom(R, P.preStopChunk, P.preStopHead);
                    L.toldCount = (int64_t)st.numBlocks;
                    L.walked = true;
                    L.exactWalk = true;
                    L.reachedEnd = !R.stopped;
                    L.runup = 0;
                    L.count = R.startSnap.column >= 0 ? (int64_t)((R.stopped ? R.endSnap.numBlocks : R.numBlocks) - R.numBlocksAtStart) : 0;
                    return;
                }
            }
            if (L.walked && !L.exactWalk && L.lastBase == baseOf[s])
                return; // (told what it was told before: the same walk)
            L.lastBase = baseOf[s];
            const int64_t told = std::max<int64_t>(0, baseOf[s] - L.runup);
            const size_t h0 = headBase[s] > RUNUP ? headBase[s] - RUNUP : 0;
            const size_t c0 = (size_t)(std::upper_bound(headBase.begin(), headBase.begin() + (std::ptrdiff_t)s + 1, h0) - headBase.begin()) - 1;
            // (the run-up's own count of blocks is only known when it has been walked: up to three times, told the count put right by
            // what the walk before found; the last one goes on whatever it finds)
            int64_t tell = told;
            size_t before = 0;
            // (HGX_MAF_SLICE_DIFF: where this walk's blocks differ from the blocks the slice's walk before found, told another count)
            std::vector<std::pair<int64_t, uint32_t>> blocksBefore;
            int64_t countBefore = -1;
            if (getenv("HGX_MAF_SLICE_DIFF") && L.R && L.R->startSnap.column >= 0) {
                for (const RunMachine::BlockLog &b : L.R->batch->blocks)
                    blocksBefore.emplace_back(b.refIndex, b.numEnts);
                countBefore = (int64_t)L.R->numBlocksAtStart;
            }
            for (int attempt = 0;; ++attempt) {
                L.R.reset(new RunMachine(*this, mafStream, refRank, ranks, (size_t)tell));
                RunMachine &R = *L.R;
                if (h0 == 0) { // (the run-up reaches the export's first column: the export's own state, not a cold one)
                    R.restoreState(initial);
                    R.numBlocks = 0;
                }
                R.logFrom = seamOf(s);
                R.stopAt = seamOf(s + 1);
                R.logging = false;
                R.expectCount = h0 == 0 || attempt == 3 ? -1 : baseOf[s];
                before = R.numBlocks;
                walkFrom(R, c0, h0 - headBase[c0]);
                if (!R.miscounted)
                    break;
                tell = std::max<int64_t>(0, tell + baseOf[s] - (int64_t)R.numBlocks);
            }
            RunMachine &R = *L.R;
            if (countBefore >= 0 && R.startSnap.column >= 0) {
                static std::mutex diffMu;
                std::lock_guard<std::mutex> lock(diffMu);
                const std::vector<RunMachine::BlockLog> &now = R.batch->blocks;
                size_t i = 0, j = 0, shown = 0;
                std::cerr << "[hgx maf]   slice " << s << ": blocks " << blocksBefore.size() << " when it began at count " << countBefore << ", " << now.size()
                          << " at count " << R.numBlocksAtStart << std::endl;
                while (i < blocksBefore.size() && j < now.size() && shown < 6) {
                    if (blocksBefore[i].first == now[j].refIndex) {
                        ++i;
                        ++j;
                        continue;
                    }
                    std::cerr << "[hgx maf]     before: block " << countBefore + (int64_t)i << " (" << (countBefore + (int64_t)i) % 1000 << " past a thousand) begins at column "
                              << blocksBefore[i].first << " with " << blocksBefore[i].second << " entries; now: block " << R.numBlocksAtStart + j << " ("
                              << (R.numBlocksAtStart + j) % 1000 << ") at " << now[j].refIndex << " with " << now[j].numEnts << std::endl;
                    ++shown;
                    if (blocksBefore[i].first < now[j].refIndex)
                        ++i;
                    else
                        ++j;
                }
            }
            L.toldCount = tell;
            L.walked = true;
            L.reachedEnd = !R.stopped;
            if (R.startSnap.column >= 0) {
                L.runup = (int64_t)(R.numBlocksAtStart - before);
                L.count = (int64_t)((R.stopped ? R.endSnap.numBlocks : R.numBlocks) - R.numBlocksAtStart);
            } else { // (no block begins inside the slice: the slice before runs through it)
                L.runup = (int64_t)(R.numBlocks - before);
                L.count = 0;
            }
        };
        const bool timing = getenv("HGX_MAF_TIMING") != nullptr;
        const auto tRound = std::chrono::steady_clock::now();
        std::vector<double> sliceWall(timing ? S : 0, 0.0), sliceCpu(timing ? S : 0, 0.0);
        auto threadCpu = []() {
            timespec ts;
            clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
            return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
        };
        auto work = [&]() {
            CpuScope cpu(g_cpuWalkNs);
            try {
                for (size_t s; (s = next.fetch_add(1)) < S;)
                    if (todo[s]) {
                        if (!timing) {
                            walkSlice(s);
                            continue;
                        }
                        const auto w0 = std::chrono::steady_clock::now();
                        const double c0 = threadCpu();
                        walkSlice(s);
                        sliceCpu[s] = threadCpu() - c0;
                        sliceWall[s] = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
                    }
            } catch (...) {
                std::lock_guard<std::mutex> lock(failureMu);
                if (!failure)
                    failure = std::current_exception();
            }
        };
        {
            std::vector<std::thread> pool;
            size_t toWalk = 0;
            for (size_t s = 1; s < S; ++s)
                toWalk += todo[s] != 0;
            const unsigned nt = (unsigned)std::min<size_t>(threads, std::max<size_t>(toWalk, 1)); // (a thread a slice that is walked)
            for (unsigned t = 1; t < nt; ++t)
                pool.emplace_back(work);
            work();
            for (std::thread &t : pool)
                t.join();
        }
        if (failure)
            std::rethrow_exception(failure);
        if (timing) { // (who the round waited for: the longest walk, and whether its thread was running all that time)
            size_t walked = 0, longest = 0;
            double sumWall = 0, sumCpu = 0;
            for (size_t s = 1; s < S; ++s)
                if (sliceWall[s] > 0) {
                    ++walked;
                    sumWall += sliceWall[s];
                    sumCpu += sliceCpu[s];
                    if (sliceWall[s] > sliceWall[longest])
                        longest = s;
                }
            std::cerr << "[hgx maf]   round " << rounds << ": " << walked << " walk(s) of " << S << " slices in "
                      << std::chrono::duration<double>(std::chrono::steady_clock::now() - tRound).count() << " s (begun "
                      << std::chrono::duration<double>(tRound - t0).count() << " s in); a walk " << (walked ? sumWall / (double)walked : 0.0) << " s on average, of which on a core "
                      << (walked ? sumCpu / (double)walked : 0.0) << " s; the longest (slice " << longest << ") " << sliceWall[longest] << " s, on a core " << sliceCpu[longest] << " s"
                      << std::endl;
        }
        const auto tCheck = std::chrono::steady_clock::now();
        // does every slice begin as the one before it ended?  (a slice whose walk ran to the export's end has everything behind it)
        settled = true;
        std::fill(todo.begin(), todo.end(), 0);
        bool ended = slice[0].reachedEnd;
        const RunMachine *prev = &first;
        size_t prevIndex = 0;               // the last slice in front that has blocks of its own
        int64_t trueCount = slice[0].count; // the blocks in front of the slice's first block, for certain
        bool prevKeysExact = true;          // the slice before was walked with that count and knew the keys it began with
        for (size_t s = 1; s < S; ++s) {
            if (ended) { // (nothing begins here: the slice in front ran through)
                slice[s].count = 0;
                continue;
            }
            RunMachine &R = *slice[s].R;
            bool agrees = false, keysExact = false;
            int64_t shift = 0;
            if (R.emptySlice && prev->stopped && prev->endSnap.column == R.emptyAt) {
                slice[s].count = 0; // (the block the slice in front ended with begins behind this slice: nothing of this one)
                slice[s].shift = 0;
                slice[s].settledEmpty = true;
                continue;
            }
            slice[s].settledEmpty = false;
            if (slice[s].exactWalk && R.startSnap.column >= 0 && prev->stopped && prev->endSnap.column == R.startSnap.column &&
                (int64_t)R.startSnap.numBlocks == trueCount) {
                agrees = keysExact = true; // (walked from the state the slice before was in: nothing to compare but that it did begin there)
            } else if (R.startSnap.column >= 0 && prev->stopped) {
                shift = trueCount - (int64_t)R.startSnap.numBlocks;
                if (shift == 0 && prevKeysExact && prev->endSnap == R.startSnap) {
                    agrees = keysExact = true; // the whole state, keys and count
                }
            }
            if (!agrees) {
                settled = false;
                if (getenv("HGX_MAF_TIMING"))
                    std::cerr << "[hgx maf]   round " << rounds << ": slice " << s << " does not begin as slice " << s - 1 << " ended (columns "
                              << (prev->stopped ? prev->endSnap.column : -1) << " / " << R.startSnap.column << ", blocks " << trueCount << " / "
                              << R.startSnap.numBlocks << ", entries " << (prev->stopped ? prev->endSnap.bRank.size() : 0) << " / " << R.startSnap.bRank.size()
                              << ", sequences seen " << (prev->stopped ? prev->endSnap.seen.size() : 0) << " / " << R.startSnap.seen.size() << ", looks at keys' entries "
                              << R.keyUses.size() << ")" << std::endl;
                // this slice and the ones behind it again, told the counts as they are known now
                for (size_t j = s; j < S; ++j)
                    todo[j] = 1;
                slice[s].exactBase = trueCount;
                frontier = s;
                frontierPrev = prevIndex;
                // the counts in front of the slices behind, as well as they can be told now: every slice's count put right by what its looks
                // at keys' entries say of the count it should have been told
                {
                    int64_t base = trueCount;
                    for (size_t j = s; j + 1 < S; ++j) {
                        const RunMachine &W = *slice[j].R;
                        int64_t count = slice[j].count;
                        if (W.startSnap.column >= 0)
                            count += W.countChangeUnderShift(base - (int64_t)W.startSnap.numBlocks);
                        slice[j].predictedCount = count;
                        slice[j].predictedShift = W.startSnap.column >= 0 ? base - (int64_t)W.startSnap.numBlocks : 0;
                        slice[j].predictedLooks = W.keyUses.size();
                        base += count;
                        slice[j + 1].guessBase = base;
                    }
                }
                break;
            }
            if (getenv("HGX_MAF_TIMING") && slice[s].predictedCount >= 0 && slice[s].predictedCount != slice[s].count)
                std::cerr << "[hgx maf]   round " << rounds << ": slice " << s << " has " << slice[s].count << " blocks, the count model said "
                          << slice[s].predictedCount << " (from a walk told a count off by " << slice[s].predictedShift << ", " << slice[s].predictedLooks
                          << " looks at keys' entries)" << std::endl;
            slice[s].predictedCount = -1;
            slice[s].shift = shift;
            slice[s].exactState = keysExact;
            prevIndex = s;
            trueCount += slice[s].count;
            prevKeysExact = keysExact;
            ended = slice[s].reachedEnd;
            prev = &R;
        }
        const auto tFlushSettled = std::chrono::steady_clock::now();
        flushSettled(settled ? S : frontier);
        if (timing)
            std::cerr << "[hgx maf]   round " << rounds << ": comparing the slices' ends " << std::chrono::duration<double>(tFlushSettled - tCheck).count()
                      << " s, handing the settled ones to the renderers " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tFlushSettled).count() << " s"
                      << std::endl;
        // the batches that arrived meanwhile come into play: behind the slices that are walked again, or — when every slice in play
        // is settled and the export goes on — after waiting for the next one, its slice walked from the state the last one stopped in
        {
            // (a settled slice whose walk ran to the export's end — it waited for the batches — has everything behind it)
            if (settled && (ended || (complete && S == there)))
                break;
            bool nowComplete = complete;
            size_t nowThere = there;
            if (settled)
                nowThere = A.waitMore(there, nowComplete);
            else
                nowThere = A.waitMore(0, nowComplete); // (does not wait: something is there)
            const size_t newS = nowComplete ? nowThere : nowThere - 1;
            if (settled) {
                // (S slices settled, more to come: the first new one is the frontier)
                frontier = S;
                frontierPrev = prevIndex;
                settled = false;
            }
            if (newS > S) {
                const size_t oldS = S;
                slice.resize(newS);
                todo.resize(newS, 1);
                for (size_t j = oldS; j < newS; ++j)
                    todo[j] = 1;
                if (frontier == oldS)
                    slice[oldS].exactBase = trueCount;
                // (the count in front of the new slices, as far as it can be told: what is known of the slices in front of them)
                int64_t base = slice[oldS - 1].exactBase >= 0 ? slice[oldS - 1].exactBase : slice[oldS - 1].guessBase >= 0 ? slice[oldS - 1].guessBase : 0;
                base += slice[oldS - 1].count;
                for (size_t j = oldS; j < newS; ++j)
                    if (slice[j].exactBase < 0)
                        slice[j].guessBase = base;
                S = newS;
            }
            there = nowThere;
            complete = nowComplete;
            headBase = A.headBases();
        }
    }
    if (getenv("HGX_MAF_TIMING"))
        std::cerr << "[hgx maf] the walk over " << S << " slices on " << threads << " threads: " << rounds << " round(s), "
                  << (settled ? "settled" : "NOT settled") << " after " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s"
                  << std::endl;
    if (roundsOut)
        *roundsOut = rounds;
    if (threadsOut)
        *threadsOut = (unsigned)std::min<size_t>(threads, S - 1);
    if (secondsOut)
        *secondsOut = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("HGX_MAF_TIMING") && getenv("HGX_MAF_SLICE_COUNTS")) {
        std::cerr << "[hgx maf]   slices' blocks (count off by):";
        for (size_t s = 0; s < S; ++s)
            std::cerr << " " << slice[s].count << "(" << slice[s].shift << (slice[s].exactWalk ? "x" : "") << ")";
        std::cerr << std::endl;
    }
    if (!settled) {
        // (cannot be: the first unsettled slice of every round is walked from the state the slice before it stopped in)
        if (flushedUpTo > 0)
            throw std::runtime_error("hal2maf: the walk over slices of the export did not settle");
        first.restoreState(initial); // the export's entries as they were: one thread's walk follows
        first.sliced = false;
        first.batch->reset();
        return false;
    }
    const RunMachine *last = &first;
    for (size_t s = 1; s < S; ++s) {
        if (slice[s].settledEmpty || !slice[s].R)
            continue;
        if (slice[s].R->startSnap.column < 0)
            continue;
        last = slice[s].R.get();
        if (slice[s].reachedEnd)
            break;
    }
    if (last != &first)
        first.restoreState(last->saveState());
    first.sliced = false;
    const size_t blocks = blocksFlushed;
    numBlocksOut = blocks ? blocks - 1 : 0; // (as the one-thread walk counts: blocks ended)
    return true;
}

void MafExport::convertSequenceRuns(std::ostream &mafStream, hgx_alignment *alignment, int genome, int seq, int64_t startPosition,
                                    int64_t length, const ColumnOptions &opt) {
    typedef RunMachine::Chunk Chunk;
    typedef RunMachine::PRow PRow;
    const GenomeTables &G = alignment->img.genomes[(size_t)genome];
    const int64_t first = startPosition + G.seqs[(size_t)seq].start;
    if (const char *e = getenv("HGX_MAF_CHUNK")) // (columns per device batch: tests cross batch ends with it)
        chunkColumns = (size_t)std::max<long long>(1, atoll(e));
    // The walk over slices of the export, side by side (walkSliced): exports of sixteen million columns or more in eight chunks or
    // more on a host of thirty-two threads or more; HGX_MAF_SLICED=1 / 0 forces / forbids it
    int64_t numChunksExpected = (length + (int64_t)chunkColumns - 1) / (int64_t)chunkColumns;
    // (a round of walks takes one slice's time when every slice has a thread: hosts of 32 threads or more)
    bool wantHeadCols = numChunksExpected >= 8 && numChunksExpected <= 512 && length >= ((int64_t)16 << 20) && hostThreads() >= 8;
    if (const char *e = getenv("HGX_MAF_SLICED"))
        wantHeadCols = atoi(e) != 0; // (1: whatever the export's size — with two batches or more; the tests and the soaks)
    // Slices are batches, and a round takes one slice's walk: where the host has a thread for each of twice as many slices, batches of
    // half the columns halve the rounds' time (full-size config 3 on the CPU replay: 53 slices settle in 12-14 rounds where 27 take
    // 9-10 — how far a round's guesses hold is a matter of columns, not of slices)
    if (wantHeadCols && !getenv("HGX_MAF_CHUNK") && chunkColumns == ((size_t)1 << 21) && hostThreads() >= 16 &&
        length >= ((int64_t)16 << 20)) {
        chunkColumns = (size_t)1 << 20;
        numChunksExpected = (length + (int64_t)chunkColumns - 1) / (int64_t)chunkColumns;
    }
    // The batches come through two stages beside the walk: the device stage (the column kernels, the copies to the host: one call at
    // a time, the next one begun as soon as this one is back, up to a few batches ahead of the walk) and the stage that describes and
    // sorts a batch's rows for the walk (several threads a batch, beside the device stage of the batch behind it).
    struct Raw {
        std::shared_ptr<Chunk> c;
        HeadRows headRows;
    };
    auto deviceStage = [&](int64_t done) {
        CpuScope cpu(g_cpuDeviceNs);
        std::unique_ptr<Raw> raw(new Raw);
        raw->c.reset(new Chunk);
        Chunk *c = raw->c.get();
        HeadRows &headRows = raw->headRows;
        c->done = done;
        c->n = std::min<int64_t>((int64_t)chunkColumns, length - done);
        const auto t0 = std::chrono::steady_clock::now();
        bool have = false;
#ifdef HGX_HOST_PROFILE
        // profiling aid of the host state machine (make hostprof-lib, not part of libhgx.so): HGX_MAF_DUMP=file records the device's
        // batches, HGX_MAF_REPLAY=file plays them back to the state machine on a machine without a GPU
        FILE *replay = mafReplayFile(), *dump = mafDumpFile();
        if (replay) {
            uint64_t hd[4];
            if (fread(hd, 8, 4, replay) != 4 || (int64_t)hd[0] != done)
                throw std::runtime_error("HGX_MAF_REPLAY: the file does not continue at this column");
            c->n = (int64_t)hd[1];
            c->head.resize((size_t)c->n);
            c->headOff.resize((size_t)hd[2]);
            headRows.resize((size_t)hd[3]);
            if (fread(c->head.data(), 1, c->head.size(), replay) != c->head.size() ||
                fread(c->headOff.data(), 4, c->headOff.size(), replay) != c->headOff.size() ||
                fread(headRows.data(), sizeof(ColumnRowHost), headRows.size(), replay) != headRows.size())
                throw std::runtime_error("HGX_MAF_REPLAY: short file");
            have = true;
        }
#endif
        while (!have) { // (a chunk's row offsets are 32-bit: very wide alignments get smaller chunks)
            try {
                columnsHeadRowsHost(alignment, genome, first + done, c->n, opt, true, c->head, c->headOff, headRows, &stats,
                                    _unique ? first : (int64_t)-1, std::max(length, _exportHint));
#ifdef HGX_HOST_PROFILE
                if (dump) {
                    const uint64_t hd[4] = {(uint64_t)done, (uint64_t)c->n, c->headOff.size(), headRows.size()};
                    fwrite(hd, 8, 4, dump);
                    fwrite(c->head.data(), 1, c->head.size(), dump);
                    fwrite(c->headOff.data(), 4, c->headOff.size(), dump);
                    fwrite(headRows.data(), sizeof(ColumnRowHost), headRows.size(), dump);
                    fflush(dump);
                }
#endif
                have = true;
            } catch (const ColumnChunkTooLarge &) {
                if (c->n <= 1)
                    throw;
                chunkColumns = (size_t)std::max<int64_t>(1, c->n / 2);
                c->n = (int64_t)chunkColumns;
            }
        }
        c->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return raw;
    };
    auto describeStage = [this, alignment, wantHeadCols](std::shared_ptr<Raw> raw) {
        CpuScope cpuAll(g_cpuDescribeNs);
        const auto t0 = std::chrono::steady_clock::now();
        Chunk *c = raw->c.get();
        const HeadRows &headRows = raw->headRows;
        c->rows.reset(static_cast<PRow *>(hostBlockTake((headRows.size() ? headRows.size() : 1) * sizeof(PRow))));
        const size_t heads = c->headOff.size() - 1;
        if (wantHeadCols) { // (the walk over slices begins in the middle of chunks)
            c->headCol.reserve(heads);
            for (int64_t i = 0; i < c->n; ++i)
                if (c->head[(size_t)i] & 1)
                    c->headCol.push_back((uint32_t)i);
        }
        auto convert = [&](size_t h0, size_t h1) {
            std::unique_ptr<CpuScope> cpu(h0 > 0 ? new CpuScope(g_cpuDescribeNs) : nullptr); // (the helpers' threads; the first share runs under cpuAll)
            for (size_t h = h0; h < h1; ++h) {
                for (size_t i = c->headOff[h]; i < c->headOff[h + 1]; ++i)
                    RunMachine::describe(alignment->img, _rank, c->rows[i], headRows[i].genome, headRows[i].pos, headRows[i].rev != 0,
                                         (uint32_t)(i - c->headOff[h]));
                RunMachine::sortColumn(c->rows.get() + c->headOff[h], c->headOff[h + 1] - c->headOff[h]);
            }
        };
        const size_t parts = describeThreads(heads);
        std::vector<std::thread> helpers;
        for (size_t t = 1; t < parts; ++t)
            helpers.emplace_back(convert, heads * t / parts, heads * (t + 1) / parts);
        convert(0, heads / parts);
        for (std::thread &t : helpers)
            t.join();
        c->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return raw->c;
    };
    // the batches in flight, in order; the thread of the device stage keeps at most `ahead` of them beyond the one being walked
    struct Pipe {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::future<std::shared_ptr<Chunk>>> ready;
        size_t taken = 0, made = 0;
        bool stop = false, done = false;
        std::exception_ptr error;
    } pipe;
    size_t ahead = getenv("HGX_MAF_AHEAD") ? (size_t)std::max(1, atoi(getenv("HGX_MAF_AHEAD"))) : 4; // (batches made and not yet taken by the walk)
    if (wantHeadCols)
        ahead = (size_t)-1; // (the walk over slices begins when every batch is there)
#ifdef HGX_HOST_PROFILE
    const bool wholeAhead = mafReplayFile() && getenv("HGX_MAF_REPLAY_AHEAD"); // (the walk by itself: every batch is there before it begins)
    if (wholeAhead)
        ahead = (size_t)-1;
#endif
    // The plain export's batches as a stream where the per-base tracks serve it (hgx_columns.hip: MafChunkStream): three launches
    // and one wait a batch, the next batch queued before this one is waited for, the rows described and sorted on the device.  A
    // batch the stream does not deliver (no room, tracks that differ from the walk) and every batch behind it come by deviceStage.
    struct StreamCloser {
        void operator()(MafChunkStream *m) const { mafChunkStreamClose(m); }
    };
    std::unique_ptr<MafChunkStream, StreamCloser> chunkStream;
    // (an export of one batch has nothing to queue ahead, and a stream's buffers, streams and page-locked arena are set up per export:
    // hgx_maf_export_multi's million-column slices went from 0.26 to 0.82 s with a stream each)
    bool streamAllowed = alignment->dev != nullptr && length > (int64_t)chunkColumns;
#ifdef HGX_HOST_PROFILE
    if (mafReplayFile() || mafDumpFile())
        streamAllowed = false; // (the recordings hold the device's rows as they were)
#endif
    std::thread deviceThread([&]() {
        try {
            int64_t nextSubmit = 0;
            if (streamAllowed) {
                std::vector<int32_t> rankBase(_rank.size(), 0);
                for (size_t g = 0; g < _rank.size(); ++g)
                    rankBase[g] = _rank[g].empty() ? 0 : _rank[g][0];
                chunkStream.reset(mafChunkStreamOpen(alignment, genome, opt, rankBase, (int64_t)chunkColumns, std::max(length, _exportHint), &stats,
                                                     _unique ? first : (int64_t)-1));
            }
            for (int64_t done = 0; done < length;) {
                {
                    std::unique_lock<std::mutex> lock(pipe.mu);
                    pipe.cv.wait(lock, [&]() { return pipe.stop || pipe.made - pipe.taken < ahead; });
                    if (pipe.stop)
                        break;
                }
                if (chunkStream) {
                    CpuScope cpu(g_cpuDeviceNs);
                    const auto t0 = std::chrono::steady_clock::now();
                    while (nextSubmit < length && mafChunkStreamInFlight(chunkStream.get()) < 3) {
                        const int64_t n = std::min<int64_t>((int64_t)chunkColumns, length - nextSubmit);
                        mafChunkStreamSubmit(chunkStream.get(), first + nextSubmit, n);
                        nextSubmit += n;
                    }
                    MafChunkOut o;
                    if (mafChunkStreamCollect(chunkStream.get(), o)) {
                        static_assert(sizeof(PRow) == sizeof(MafChunkRow), "the stream's rows are the walk's");
                        std::shared_ptr<Chunk> c(new Chunk);
                        c->done = done;
                        c->n = o.n;
                        c->head.swap(o.head);
                        c->headOff.swap(o.headOff);
                        if (wantHeadCols)
                            c->headCol.swap(o.headCol);
                        c->rows.reset(reinterpret_cast<PRow *>(o.rows));
                        c->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        done += c->n;
                        std::promise<std::shared_ptr<Chunk>> ready;
                        ready.set_value(c);
                        std::lock_guard<std::mutex> lock(pipe.mu);
                        pipe.ready.push_back(ready.get_future());
                        ++pipe.made;
                        pipe.cv.notify_all();
                        continue;
                    }
                    chunkStream.reset(); // (waits for what it had queued; this batch and the rest by deviceStage)
                }
                std::shared_ptr<Raw> raw(deviceStage(done).release());
                done += raw->c->n;
                std::future<std::shared_ptr<Chunk>> f = std::async(std::launch::async, describeStage, raw);
                std::lock_guard<std::mutex> lock(pipe.mu);
                pipe.ready.push_back(std::move(f));
                ++pipe.made;
                pipe.cv.notify_all();
            }
        } catch (...) {
            std::lock_guard<std::mutex> lock(pipe.mu);
            pipe.error = std::current_exception();
        }
        std::lock_guard<std::mutex> lock(pipe.mu);
        pipe.done = true;
        pipe.cv.notify_all();
    });
    struct Join { // (whatever ends the walk ends the device stage's thread too)
        Pipe &p;
        std::thread &t;
        ~Join() {
            {
                std::lock_guard<std::mutex> lock(p.mu);
                p.stop = true;
                p.cv.notify_all();
            }
            t.join();
            for (auto &f : p.ready) // (stages still describing their batch)
                if (f.valid())
                    f.wait();
        }
    } join{pipe, deviceThread};
    auto nextChunk = [&]() {
        std::future<std::shared_ptr<Chunk>> f;
        {
            std::unique_lock<std::mutex> lock(pipe.mu);
            pipe.cv.wait(lock, [&]() { return !pipe.ready.empty() || pipe.done; });
            if (pipe.ready.empty()) {
                if (pipe.error)
                    std::rethrow_exception(pipe.error);
                throw std::runtime_error("hal2maf: the batches ended before the columns did");
            }
            f = std::move(pipe.ready.front());
            pipe.ready.pop_front();
            ++pipe.taken;
            pipe.cv.notify_all();
        }
        return f.get();
    };
    double fetchSeconds = 0, waitSeconds = 0;
    size_t numHeads = 0, numBlocks = 0;
#ifdef HGX_HOST_PROFILE
    if (wholeAhead) {
        std::unique_lock<std::mutex> lock(pipe.mu);
        pipe.cv.wait(lock, [&]() { return pipe.done; });
        for (auto &f : pipe.ready)
            f.wait();
    }
#endif
    const auto tStart = std::chrono::steady_clock::now();
    Arrivals arrivals; // (the walk over slices: every batch, kept until the text is written)
    bool slicedDone = false;
    int slicedRounds = 0;
    unsigned slicedThreads = 0;
    double slicedFetched = 0, slicedWalk = 0;
    if (wantHeadCols) {
        // the batches are handed to the walk as they arrive (a thread of its own takes them from the stages); the walk's rounds and
        // the rendering of what they settle run beside the device stage
        arrivals.since = tStart;
        struct Feeder {
            std::thread t;
            ~Feeder() {
                if (t.joinable())
                    t.join();
            }
        } feeder;
        const long feedDelayUs = getenv("HGX_MAF_FEED_DELAY_US") ? atol(getenv("HGX_MAF_FEED_DELAY_US")) : 0;
        feeder.t = std::thread([&]() {
            try {
                for (int64_t done = 0; done < length;) {
                    const auto tw = std::chrono::steady_clock::now();
                    std::shared_ptr<Chunk> c = nextChunk();
                    waitSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
                    fetchSeconds += c->seconds;
                    numHeads += c->headOff.size() - 1;
                    done += c->n;
                    if (feedDelayUs > 0) // (the tests: batches that arrive while the rounds run)
                        std::this_thread::sleep_for(std::chrono::microseconds(feedDelayUs));
                    arrivals.push(std::move(c));
                }
                arrivals.end();
            } catch (...) {
                arrivals.end(std::current_exception());
            }
        });
        try {
            slicedDone = walkSliced(mafStream, arrivals, _rank[(size_t)genome][(size_t)seq], startPosition, numBlocks, &slicedRounds, &slicedThreads,
                                    &slicedWalk);
        } catch (...) {
            // (the feeding thread ends by itself: the device stage goes on to the export's end or to its own failure; Join below ends
            // the device stage's thread when this frame is left, the feeder's nextChunk then throws and the feeder ends)
            {
                std::lock_guard<std::mutex> lock(pipe.mu);
                pipe.stop = true;
                pipe.cv.notify_all();
            }
            feeder.t.join();
            // the batches settled so far are being rendered from rows that `arrivals` and the slices' machines keep alive: no thread of
            // theirs may outlive this frame (their own failures are secondary to the one on its way out)
            try {
                waitPendingWrite();
            } catch (...) {
            }
            throw;
        }
        feeder.t.join();
        arrivals.waitComplete(); // (rethrows the feeding thread's failure)
        slicedFetched = arrivals.completeAt;
        if (getenv("HGX_MAF_TIMING"))
            std::cerr << "[hgx maf] columns " << length << " heads " << numHeads << " blocks " << numBlocks << ": the walk over " << arrivals.chunks.size()
                      << " slices " << (slicedDone ? "" : "did not begin (one batch); one thread's walk instead ") << "after "
                      << std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count() << " s, the last batch there after "
                      << slicedFetched << " s (fetches " << fetchSeconds << " s, device " << stats.rows_ms + stats.depth_ms << " ms)" << std::endl;
    }
    const std::vector<std::shared_ptr<Chunk>> &all = arrivals.chunks; // (complete by now)
    if (!slicedDone) {
        RunMachine R(*this, mafStream, _rank[(size_t)genome][(size_t)seq]);
        size_t nextOfAll = 0;
        for (int64_t done = 0; done < length;) {
            const auto tw = std::chrono::steady_clock::now();
            std::shared_ptr<Chunk> c = wantHeadCols ? all[nextOfAll++] : nextChunk();
            waitSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
            const int64_t n = c->n;
            if (!wantHeadCols) {
                fetchSeconds += c->seconds;
                numHeads += c->headOff.size() - 1;
            }
            {
                CpuScope cpu(g_cpuWalkNs);
                R.walkChunk(c, 0, startPosition);
            }
            done += n;
        }
        if (R.appendCount > 0)
            R.endBlock();
        R.flush();
        numBlocks = R.numBlocks;
        if (getenv("HGX_MAF_TIMING"))
            std::cerr << "[hgx maf] columns " << length << " heads " << numHeads << " blocks " << numBlocks << " state machine + waits "
                      << std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count() << " s, of which waiting for the device "
                      << waitSeconds << " s (fetches " << fetchSeconds << " s, device " << stats.rows_ms + stats.depth_ms << " ms), for the rendering of the batch before "
                      << g_mafRenderWait << " s, handing batches over " << g_mafFlush << " s" << std::endl;
        g_mafRenderWait = 0;
        g_mafFlush = 0;
    }
    waitPendingWrite();
    mafStream.flush();
    if (getenv("HGX_MAF_TIMING")) {
        std::cerr << "[hgx maf] CPU seconds of the export's stages (all their threads; " << hostThreads() << " host threads to go by): device stage's thread "
                  << 1e-9 * (double)g_cpuDeviceNs.exchange(0) << ", describing and sorting rows " << 1e-9 * (double)g_cpuDescribeNs.exchange(0) << ", walks "
                  << 1e-9 * (double)g_cpuWalkNs.exchange(0) << ", rendering " << 1e-9 * (double)g_cpuRenderNs.exchange(0) << ", copying the text into place "
                  << 1e-9 * (double)g_cpuCopyNs.exchange(0) << "; the export took " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count()
                  << " s" << std::endl;
    } else {
        g_cpuDeviceNs = 0;
        g_cpuDescribeNs = 0;
        g_cpuWalkNs = 0;
        g_cpuRenderNs = 0;
        g_cpuCopyNs = 0;
    }
    {
        char buf[640];
        snprintf(buf, sizeof buf,
                 "{\"columns\": %lld, \"heads\": %zu, \"blocks\": %zu, \"batches\": %zu, \"walk\": \"%s\", \"slices\": %zu, \"rounds\": %d, "
                 "\"walk_threads\": %u, \"seconds_until_the_batches_were_there\": %.4f, \"seconds_of_the_rounds\": %.4f, \"seconds\": %.4f, "
                 "\"seconds_waiting_for_the_device\": %.4f, \"device_stage_seconds\": %.4f}",
                 (long long)length, numHeads, numBlocks, wantHeadCols ? all.size() : (size_t)numChunksExpected,
                 slicedDone ? "slices of the export side by side" : "one thread", slicedDone ? all.size() : (size_t)0, slicedRounds, slicedThreads,
                 slicedFetched, slicedWalk, std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count(), waitSeconds, fetchSeconds);
        std::lock_guard<std::mutex> lock(lastExport().mu);
        lastExport().json = buf;
    }
#ifdef HGX_HOST_PROFILE
    if (getenv("HGX_MAF_TIMING")) {
        static const char *what[] = {"", "initBlock", "canAppend", "pair + place"};
        for (int i = 1; i < 4; ++i)
            std::cerr << "[hgx maf]   " << what[i] << " " << (double)g_mafTicks[i] / 1e6 << " Mticks" << std::endl;
        std::cerr << "[hgx maf]   per block: entries " << (double)g_mafTicks[8] / numBlocks << " rows " << (double)g_mafTicks[9] / numBlocks << " keys "
                  << (double)g_mafTicks[10] / numBlocks << std::endl;
    }
#endif
    if (getenv("HGX_MAF_TIMING"))
        std::cerr << "[hgx maf] written after " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count() << " s" << std::endl;
}

// halMafExport.cpp:25-88
void MafExport::convertSequence(std::ostream &mafStream, hgx_alignment *alignment, int genome, int seq, int64_t startPosition,
                                int64_t length, const std::set<int> &targets) {
    if (_al != alignment) {
        _al = alignment;
        buildRanks();
    }
    const GenomeTables &G = alignment->img.genomes[(size_t)genome];
    const SeqInfo &S = G.seqs[(size_t)seq];
    if (startPosition >= S.length || startPosition + length > S.length)
        throw std::runtime_error("Invalid range specified for convertGenome");
    if (length == 0)
        length = S.length - startPosition;
    if (length == 0)
        throw std::runtime_error("Cannot convert zero length sequence");
    // writeHeader (halMafExport.cpp:15-23) as the reference has it: whenever the stream's position is not past its start —
    // once for a file or a string stream, before every sequence on a stream that cannot tell its position (a pipe: tellp()
    // is -1, and hal2maf to stdout repeats the header per sequence and per --refTargets interval)
    if (!_append && mafStream.tellp() <= std::streampos(0)) {
        mafStream << "##maf version=1 scoring=N/A\n"
                  << "# hal " << alignment->img.newick << std::endl
                  << std::endl;
    }
    ColumnOptions opt;
    opt.noDupes = _noDupes;
    opt.noAncestors = _noAncestors;
    opt.onlyOrthologs = _onlyOrthologs;
    opt.targets.assign(targets.begin(), targets.end());
    const Key refKey{_rank[(size_t)genome][(size_t)seq], genome, seq};

    // --unique alone: which columns the iterator walks and writes is decided on the device, column by column (hgx_column_kernels.hpp:
    // k_column_unique_count), and the written ones go through the run-compressed path; HGX_UNIQUE_REPLAY=1 keeps the replay of the
    // visit cache on the host (below) as a cross-check
    if (_unique && _maxRefGap == 0 && !_printTree && !getenv("HGX_UNIQUE_REPLAY") && !getenv("HGX_MAF_PER_COLUMN") && !getenv("HGX_MAF_MAP_STATE")) {
        convertSequenceRuns(mafStream, alignment, genome, seq, startPosition, length, opt);
        return;
    }
    // (the iterator whose state is sequential — a visit cache, a stack of ranges — is replayed on the host over the device's columns)
    if (_maxRefGap > 0 || _unique) {
        convertSequenceGapped(mafStream, alignment, genome, seq, startPosition + S.start, startPosition + S.start + length - 1, opt);
        return;
    }
    ColumnMap colMap; // keys persist between columns like ColumnIterator::_colMap (resetColMap only empties the vectors)
    std::vector<uint64_t> off;
    std::vector<ColumnRowHost> rows;
    size_t appendCount = 0, numBlocks = 0;
    const int64_t first = startPosition + S.start, last = first + length - 1;

    // The GPU delivers every column of a chunk; the iterator state that is sequential by definition in the reference —
    // the visit cache of --unique (halColumnIterator.cpp:749-819) and the MAF block state — is replayed here.
    std::map<int64_t, int64_t> cache; // PositionCache of the reference genome (halPositionCache.cpp): last -> first
    auto cacheFind = [&](int64_t pos) {
        auto i = cache.lower_bound(pos);
        return i != cache.end() && i->second <= pos;
    };
    auto cacheInsert = [&](int64_t pos) {
        if (cacheFind(pos))
            return false;
        int64_t lo = pos, hi = pos;
        auto right = cache.lower_bound(pos);
        if (right != cache.end() && right->second == pos + 1) {
            hi = right->first;
            cache.erase(right);
        }
        auto left = cache.find(pos - 1);
        if (left != cache.end()) {
            lo = left->second;
            cache.erase(left);
        }
        cache[hi] = lo;
        return true;
    };
    int64_t chunkFirst = 0, chunkCount = 0; // genome coordinates covered by off/rows
    auto fetch = [&](int64_t pos) {
        if (pos < chunkFirst || pos >= chunkFirst + chunkCount) {
            chunkFirst = pos;
            chunkCount = std::min<int64_t>((int64_t)chunkColumns, last - pos + 1);
            for (auto &kv : colMap)
                kv.second.clear(); // they point into the buffer about to be replaced
#ifdef HGX_HOST_PROFILE
            // (the profiling build's recording / playback of the device's batches, as in convertSequenceRuns: HGX_MAF_DUMP / HGX_MAF_REPLAY)
            FILE *replay = mafReplayFile(), *dump = mafDumpFile();
            if (replay) {
                uint64_t hd[4];
                if (fread(hd, 8, 4, replay) != 4 || (int64_t)hd[0] != chunkFirst || (int64_t)hd[1] != chunkCount)
                    throw std::runtime_error("HGX_MAF_REPLAY: the file does not continue with these columns");
                off.resize((size_t)hd[2]);
                rows.resize((size_t)hd[3]);
                if (fread(off.data(), 8, off.size(), replay) != off.size() || fread(rows.data(), sizeof(ColumnRowHost), rows.size(), replay) != rows.size())
                    throw std::runtime_error("HGX_MAF_REPLAY: short file");
                return (size_t)(pos - chunkFirst);
            }
#endif
            columnsRowsHost(alignment, genome, chunkFirst, chunkCount, opt, true, off, rows, &stats);
#ifdef HGX_HOST_PROFILE
            if (dump) {
                const uint64_t hd[4] = {(uint64_t)chunkFirst, (uint64_t)chunkCount, off.size(), rows.size()};
                fwrite(hd, 8, 4, dump);
                fwrite(off.data(), 8, off.size(), dump);
                fwrite(rows.data(), sizeof(ColumnRowHost), rows.size(), dump);
                fflush(dump);
            }
#endif
        }
        return (size_t)(pos - chunkFirst);
    };

    int64_t index = first;         // ColumnIterator's stack entry (_index), genome coordinates
    int64_t leftmostRefPos = first; // _leftmostRefPos
    int64_t prevRefIndex = 0;       // getReferenceSequencePosition()
    // ColumnIterator::toRight (halColumnIterator.cpp:65-144), one-entry stack
    auto toRight = [&]() {
        prevRefIndex = index - S.start;
        if (index < first || index > last)
            return;
        bool brk;
        do {
            if (_unique)
                while (cacheFind(index) && index <= last) // nextFreeIndex
                    ++index;
            if (index < first || index > last)
                return; // the column map keeps whatever the last walk left in it
            // recursiveUpdate + colMapInsert (:246-355, :766-819)
            const size_t i = fetch(index);
            for (auto &kv : colMap)
                kv.second.clear();
            brk = false;
            leftmostRefPos = index;
            for (uint64_t r = off[i]; r < off[i + 1]; ++r) {
                const ColumnRowHost &row = rows[r];
                bool found = false;
                if (_unique && row.genome == genome)
                    found = first < row.pos ? !cacheInsert(row.pos) : cacheFind(row.pos);
                if (!found)
                    colMap[keyOf(row.genome, row.pos)].push_back(&row);
                if (row.genome == genome)
                    leftmostRefPos = std::min(leftmostRefPos, row.pos);
                if (found) {
                    brk = true; // the walk stops at the first base already visited
                    break;
                }
            }
            ++index;
        } while (brk);
        // "clean stack again" (:125-130): the index moves on to the next base not visited yet before the caller asks
        // lastColumn() — a range whose last bases were all seen in earlier columns ends here, not with one more column
        if (_unique)
            while (cacheFind(index) && index <= last)
                ++index;
    };
    auto canonicalOnRef = [&]() { return leftmostRefPos >= first && leftmostRefPos <= last; }; // :210-214

    // ---- fast path (no --unique): run-compressed columns ----
    // Inside a run every column has the rows of its left neighbour advanced by one base, so canAppendColumn can only
    // fail on the block-length limit and appendColumn only appends one character per entry; the run is therefore
    // appended in bulk, falling back to single columns at sequence ends and block-length breaks.
    if (!_unique && !_printTree && !getenv("HGX_MAF_PER_COLUMN") && !getenv("HGX_MAF_MAP_STATE")) {
        convertSequenceRuns(mafStream, alignment, genome, seq, startPosition, length, opt);
        return;
    }
    if (!_unique && !_printTree && !getenv("HGX_MAF_PER_COLUMN")) { // (the same path on MafBlock's own containers: kept as a cross-check, HGX_MAF_MAP_STATE=1)
        struct SegModeGuard {
            bool &f;
            explicit SegModeGuard(bool &x) : f(x) { f = true; }
            ~SegModeGuard() { f = false; }
        } segModeGuard(_segMode);
        _snapBlocks.clear();
        _snapRows.clear();
        _snapSegs.clear();
        double fetchSeconds = 0;
        size_t numHeads = 0;
        std::vector<uint8_t> head;
        std::vector<uint32_t> headOff;
        HeadRows headRows;
        std::vector<ColumnRowHost> curRows;
        struct Pair {
            Entry *e;
            ColumnRowHost *row; // null: entry gets a gap
        };
        std::vector<Pair> pairs;
        auto rebuildColMap = [&]() {
            for (auto &kv : colMap)
                kv.second.clear();
            for (ColumnRowHost &r : curRows)
                colMap[keyOf(r.genome, r.pos)].push_back(&r);
        };
        auto buildPairs = [&]() { // the pairing appendColumn performs (halMafBlock.cpp:370-395)
            pairs.clear();
            Entries::iterator e = _entries.begin();
            for (auto c = colMap.begin(); c != colMap.end(); ++c)
                for (const ColumnRowHost *row : c->second) {
                    while (e != _entries.end() && e->first.rank != c->first.rank) {
                        pairs.push_back(Pair{e->second, nullptr});
                        ++e;
                    }
                    pairs.push_back(Pair{e->second, const_cast<ColumnRowHost *>(row)});
                    ++e;
                }
            for (; e != _entries.end(); ++e)
                pairs.push_back(Pair{e->second, nullptr});
        };
        double tCan = 0, tPrint = 0, tInit = 0, tAppend = 0, tMap = 0, tPairs = 0, tBulk = 0;
        const bool timing = getenv("HGX_MAF_TIMING") != nullptr;
        auto now = [&]() { return timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point(); };
        auto since = [&](std::chrono::steady_clock::time_point t0) {
            return timing ? std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() : 0.0;
        };
        auto stepColumn = [&](int64_t refPos) { // one column through the reference's per-column logic
            if (appendCount == 0) {
                initBlock(colMap, refKey, refPos);
            } else {
                auto t0 = now();
                const bool can = canAppendColumn(colMap);
                tCan += since(t0);
                if (!can) {
                    if (numBlocks++ % 1000 == 0)
                        for (auto it = colMap.begin(); it != colMap.end();)
                            it = it->second.empty() ? colMap.erase(it) : std::next(it);
                    if (_keepEmptyRefBlocks || !referenceIsAllGaps()) {
                        t0 = now();
                        snapshotBlock();
                        if (_snapBlocks.size() >= 32768)
                            flushSnapshots(mafStream);
                        tPrint += since(t0);
                    }
                    t0 = now();
                    initBlock(colMap, refKey, refPos);
                    tInit += since(t0);
                }
            }
            auto t1 = now();
            appendColumn(colMap);
            tAppend += since(t1);
            ++appendCount;
        };
        for (int64_t done = 0; done < length;) {
            int64_t n = std::min<int64_t>((int64_t)chunkColumns, length - done);
            const auto tFetch0 = std::chrono::steady_clock::now();
            for (;;) { // (a chunk's row offsets are 32-bit: very wide alignments get smaller chunks)
                try {
                    columnsHeadRowsHost(alignment, genome, first + done, n, opt, true, head, headOff, headRows, &stats, -1, length);
                    break;
                } catch (const ColumnChunkTooLarge &) {
                    if (n <= 1)
                        throw;
                    chunkColumns = (size_t)std::max<int64_t>(1, n / 2);
                    n = (int64_t)chunkColumns;
                }
            }
            fetchSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tFetch0).count();
            numHeads += headOff.size() - 1;
            size_t hk = 0;
            for (int64_t i = 0; i < n;) {
                // head column i: its rows come from the GPU
                auto tm0 = now();
                curRows.assign(headRows.begin() + headOff[hk], headRows.begin() + headOff[hk + 1]);
                ++hk;
                rebuildColMap();
                tMap += since(tm0);
                stepColumn(startPosition + done + i);
                int64_t run = 0; // continuation columns that follow
                while (i + 1 + run < n && !head[(size_t)(i + 1 + run)])
                    ++run;
                int64_t col = i + 1;
                while (run > 0) {
                    auto tp0 = now();
                    buildPairs();
                    tPairs += since(tp0);
                    auto tb0 = now();
                    int64_t t = run;
                    for (const Pair &pr : pairs) {
                        if (!pr.row)
                            continue;
                        t = std::min(t, _maxBlockLength - pr.e->length); // canAppendColumn: length >= maxLength breaks
                        const GenomeTables &RG = alignment->img.genomes[(size_t)pr.row->genome];
                        const SeqInfo &RS = RG.seqs.size() == 1 ? RG.seqs[0] : RG.seqs[(size_t)RG.seqIndexBySite(pr.row->pos)];
                        t = std::min(t, pr.row->rev ? pr.row->pos - RS.start : RS.start + RS.length - 1 - pr.row->pos);
                    }
                    if (t > 0) {
                        for (const Pair &pr : pairs) {
                            if (!pr.row) {
                                appendRun(pr.e, nullptr, 0, t);
                                continue;
                            }
                            ColumnRowHost &r = *pr.row;
                            appendRun(pr.e, &r, r.rev ? r.pos - 1 : r.pos + 1, t); // the text is rendered when the block is printed
                            pr.e->length += t;
                            r.pos += r.rev ? -t : t;
                        }
                        appendCount += (size_t)t;
                        run -= t;
                        col += t;
                    }
                    tBulk += since(tb0);
                    if (run > 0) { // a block-length break or a sequence end: one ordinary column
                        for (ColumnRowHost &r : curRows)
                            r.pos += r.rev ? -1 : 1; // (the base itself is not needed: rows are kept as runs)
                        rebuildColMap();
                        stepColumn(startPosition + done + col);
                        --run;
                        ++col;
                    }
                }
                i = col;
            }
            done += n;
        }
        if (appendCount > 0 && (_keepEmptyRefBlocks || !referenceIsAllGaps()))
            snapshotBlock();
        {
            auto t0 = now();
            flushSnapshots(mafStream);
            waitPendingWrite();
            mafStream.flush();
            tPrint += since(t0);
        }
        if (getenv("HGX_MAF_TIMING"))
            std::cerr << "[hgx maf] columns " << length << " heads " << numHeads << " blocks " << numBlocks << " fetch(GPU+copy) "
                      << fetchSeconds << " s, device " << stats.rows_ms + stats.depth_ms << " ms; host: colMap " << tMap << " canAppend " << tCan
                      << " print " << tPrint << " initBlock " << tInit << " appendColumn " << tAppend << " pairs " << tPairs << " bulk " << tBulk
                      << " s" << std::endl;
        return;
    }

    toRight(); // the constructor's first step
    if (!_unique || canonicalOnRef()) {
        initBlock(colMap, refKey, prevRefIndex);
        appendColumn(colMap);
        ++appendCount;
    }
    while (!(index > last)) { // lastColumn()
        toRight();
        if (!_unique || canonicalOnRef()) {
            if (appendCount == 0)
                initBlock(colMap, refKey, prevRefIndex);
            if (!canAppendColumn(colMap)) {
                if (numBlocks++ % 1000 == 0) { // ColumnIterator::defragment (halColumnIterator.cpp:193-208)
                    for (auto it = colMap.begin(); it != colMap.end();)
                        it = it->second.empty() ? colMap.erase(it) : std::next(it);
                }
                if (appendCount > 0 && (_keepEmptyRefBlocks || !referenceIsAllGaps())) {
                    printBlock(mafStream);
                    mafStream << '\n';
                }
                initBlock(colMap, refKey, prevRefIndex);
            }
            appendColumn(colMap);
            ++appendCount;
        }
    }
    if (appendCount > 0 && (_keepEmptyRefBlocks || !referenceIsAllGaps())) {
        printBlock(mafStream);
        mafStream << std::endl;
    }
}

// ---------------------------------------------------------------------------------------------
// ColumnIterator with maxInsertLength > 0.  The device walks columns and reports what their walks meet (hgx_gap_kernels.hpp);
// the iterator's sequential state — the stack of ranges (api/inc/halColumnIteratorStack.h), the visit cache of the reference
// genome and of every genome a range lies in (halColumnIterator.cpp:749-819), the columns abandoned at a base seen before —
// is replayed here, in the reference's order.
namespace {

struct PositionCache { // api/impl/halPositionCache.cpp:12-78: positions as merged intervals, first -> last
    std::map<int64_t, int64_t> set;
    // (a walk along a sequence asks for, and adds, the position behind the one before: the interval touched last is tried first)
    mutable std::map<int64_t, int64_t>::iterator hint = set.end();
    PositionCache() = default;
    PositionCache(const PositionCache &o) : set(o.set) { hint = set.end(); }
    PositionCache(PositionCache &&o) noexcept : set(std::move(o.set)) { hint = set.end(); o.hint = o.set.end(); }
    PositionCache &operator=(PositionCache o) {
        set.swap(o.set);
        hint = set.end();
        return *this;
    }
    bool find(int64_t pos) const {
        if (hint != set.end() && hint->first <= pos && pos <= hint->second)
            return true;
        std::map<int64_t, int64_t> &m = const_cast<std::map<int64_t, int64_t> &>(set);
        auto i = m.upper_bound(pos); // the first interval that begins behind pos
        if (i == m.begin())
            return false;
        --i;
        if (i->second < pos)
            return false;
        hint = i;
        return true;
    }
    bool insert(int64_t pos) { // false: already there
        if (hint != set.end() && hint->first <= pos) {
            if (pos <= hint->second)
                return false;
            if (pos == hint->second + 1) { // the common case: one more position at the end of the interval touched last
                auto next = std::next(hint);
                if (next == set.end() || next->first > pos + 1) {
                    hint->second = pos;
                    return true;
                }
            }
        }
        auto right = set.upper_bound(pos); // the first interval that begins behind pos
        auto left = right;
        bool haveLeft = false;
        if (left != set.begin()) {
            --left;
            haveLeft = true;
            if (left->second >= pos) {
                hint = left;
                return false;
            }
        }
        const bool joinsLeft = haveLeft && left->second == pos - 1, joinsRight = right != set.end() && right->first == pos + 1;
        if (joinsLeft && joinsRight) {
            left->second = right->second;
            set.erase(right);
            hint = left;
        } else if (joinsLeft) {
            left->second = pos;
            hint = left;
        } else if (joinsRight) {
            const int64_t hi = right->second;
            set.erase(right);
            hint = set.emplace(pos, hi).first;
        } else {
            hint = set.emplace(pos, pos).first;
        }
        return true;
    }
};

struct StackEntry { // ColumnIteratorStack::Entry (halColumnIteratorStack.h:47-107)
    int g = -1;
    int64_t firstIndex = 0, index = 0, lastIndex = 0, cumulativeSize = 0;
    bool reversed = false;
    bool inBounds() const { return index >= firstIndex && index <= lastIndex; }
};

// ColumnIteratorStack::push on one of the indel stacks (:112-121)
void pushEntry(std::vector<StackEntry> &st, int g, int64_t index, int64_t lastIndex, bool reversed) {
    StackEntry e;
    e.g = g;
    e.cumulativeSize = st.empty() ? 0 : st.back().cumulativeSize + lastIndex - index + 1;
    e.firstIndex = index;
    e.index = reversed ? lastIndex : index;
    e.lastIndex = lastIndex;
    e.reversed = reversed;
    st.push_back(e);
}

struct ColKey {
    int64_t pos;
    int32_t genome, reversed;
    bool operator<(const ColKey &o) const {
        if (genome != o.genome)
            return genome < o.genome;
        if (reversed != o.reversed)
            return reversed < o.reversed;
        return pos < o.pos;
    }
};

// the columns the device has delivered: batches of (asks, row offsets, rows)
struct GapColumns {
    hgx_alignment *al;
    int refGenome;
    const ColumnOptions &opt;
    int64_t maxInsert;
    ColumnStats *stats;
    struct Batch {
        std::vector<GapAskHost> asks;
        std::vector<uint64_t> off;
        std::vector<ColumnRowHost> rows;
    };
    std::deque<Batch> batches, previous; // previous: the chunk before (the column map may still point into it)
    std::map<ColKey, std::pair<uint32_t, uint32_t>> where; // (the columns of the ranges; the reference columns are found by their index)
    int64_t baseFirst = 0, baseCount = 0;
    GapColumns(hgx_alignment *a, int ref, const ColumnOptions &o, int64_t mi, ColumnStats *st) : al(a), refGenome(ref), opt(o), maxInsert(mi), stats(st) {}

    void run(std::vector<GapAskHost> &&asks, bool index = true) {
        batches.emplace_back();
        Batch &B = batches.back();
        B.asks = std::move(asks);
        const bool events = maxInsert > 0;
#ifdef HGX_HOST_PROFILE
        // (the profiling build's recording / playback of the device's batches: HGX_MAF_DUMP / HGX_MAF_REPLAY)
        if (FILE *replay = mafReplayFile()) {
            uint64_t hd[4];
            if (fread(hd, 8, 4, replay) != 4 || hd[0] != 0x4741505Full || hd[1] != B.asks.size())
                throw std::runtime_error("HGX_MAF_REPLAY: the file does not continue with this batch of columns");
            B.off.resize((size_t)hd[2]);
            B.rows.resize((size_t)hd[3]);
            if (fread(B.off.data(), 8, B.off.size(), replay) != B.off.size() ||
                fread(B.rows.data(), sizeof(ColumnRowHost), B.rows.size(), replay) != B.rows.size())
                throw std::runtime_error("HGX_MAF_REPLAY: short file");
        } else {
            columnsGapRowsHost(al, refGenome, B.asks, opt, true, B.off, B.rows, stats, events);
            if (FILE *dump = mafDumpFile()) {
                const uint64_t hd[4] = {0x4741505Full, B.asks.size(), B.off.size(), B.rows.size()};
                fwrite(hd, 8, 4, dump);
                fwrite(B.off.data(), 8, B.off.size(), dump);
                fwrite(B.rows.data(), sizeof(ColumnRowHost), B.rows.size(), dump);
                fflush(dump);
            }
        }
#else
        columnsGapRowsHost(al, refGenome, B.asks, opt, true, B.off, B.rows, stats, events);
#endif
        if (index)
            for (size_t i = 0; i < B.asks.size(); ++i)
                where[ColKey{B.asks[i].pos, B.asks[i].genome, B.asks[i].reversed}] = {(uint32_t)batches.size() - 1, (uint32_t)i};
    }
    bool usable(const ColumnRowHost &head, const ColumnRowHost &tail) const { // a range the iterator can walk (and may accept)
        const int64_t first = head.pos, last = tail.pos;
        return first >= 0 && last >= first && last < al->img.genomes[(size_t)head.genome].totalLength && last - first + 1 <= maxInsert;
    }
    // the reference columns [first, first + count) and, level by level, every column of every range their walks (and the walks of
    // those columns ...) could push: a superset of what the replay will ask for
    // (without ranges to follow — maxInsert == 0 — the chunk behind this one is walked by the device and copied while this one is
    // replayed: ahead = the columns it would have if the replay goes straight on, 0 for none)
    std::future<std::unique_ptr<Batch>> coming;
    int64_t comingFirst = 0, comingCount = 0;
    std::unique_ptr<Batch> walk(int64_t first, int64_t count) const {
        std::unique_ptr<Batch> B(new Batch);
        B->asks.resize((size_t)count);
        for (int64_t i = 0; i < count; ++i)
            B->asks[(size_t)i] = GapAskHost{first + i, refGenome, 0};
        columnsGapRowsHost(al, refGenome, B->asks, opt, true, B->off, B->rows, stats, false);
        return B;
    }
    ~GapColumns() {
        if (coming.valid())
            coming.wait();
    }
    // first / count: in: the columns asked for; out: the columns delivered (the chunk foreseen, when it holds the first one asked for)
    void prefetch(int64_t &first, int64_t &count, int64_t lastIndex) {
        MAF_TICK(6);
        previous.swap(batches);
        batches.clear();
        where.clear();
        bool async = maxInsert == 0;
#ifdef HGX_HOST_PROFILE
        async = async && !mafReplayFile() && !mafDumpFile(); // (a recording holds the batches in the order the replay asks for them)
#endif
        if (async && coming.valid() && first >= comingFirst && first < comingFirst + comingCount) {
            batches.push_back(std::move(*coming.get()));
            first = comingFirst;
            count = comingCount;
        } else {
            if (coming.valid())
                coming.get(); // (the replay jumped: not the chunk that was foreseen)
            std::vector<GapAskHost> asks((size_t)count);
            for (int64_t i = 0; i < count; ++i)
                asks[(size_t)i] = GapAskHost{first + i, refGenome, 0};
            run(std::move(asks), false);
        }
        const int64_t ahead = std::min<int64_t>(count, lastIndex - (first + count) + 1);
        if (async && ahead > 0) {
            comingFirst = first + count;
            comingCount = ahead;
            coming = std::async(std::launch::async, [this]() { return walk(comingFirst, comingCount); });
        }
        baseFirst = first;
        baseCount = count;
        for (size_t level = 0; maxInsert > 0 && level < batches.size(); ++level) { // (a batch may add the next one)
            const Batch &B = batches[level];
            std::vector<GapAskHost> next;
            std::set<ColKey> seen;
            for (size_t r = 0; r + 1 < B.rows.size(); ++r) {
                const int kind = B.rows[r]._pad[0] & 7;
                if ((kind != 2 && kind != 3) || !usable(B.rows[r], B.rows[r + 1]))
                    continue;
                for (int64_t p = B.rows[r].pos; p <= B.rows[r + 1].pos; ++p) {
                    const ColKey k{p, B.rows[r].genome, B.rows[r].rev ? 1 : 0};
                    const bool base = k.genome == refGenome && !k.reversed && p >= baseFirst && p < baseFirst + baseCount;
                    if (!base && !where.count(k) && seen.insert(k).second)
                        next.push_back(GapAskHost{p, k.genome, k.reversed});
                }
            }
            if (!next.empty())
                run(std::move(next));
        }
    }
    // rows of the column that starts from (genome, pos, reversed): [begin, end)
    void column(int g, int64_t pos, bool reversed, const ColumnRowHost *&begin, const ColumnRowHost *&end) {
        if (g == refGenome && !reversed && pos >= baseFirst && pos < baseFirst + baseCount && !batches.empty()) {
            const Batch &B = batches[0];
            begin = B.rows.data() + B.off[(size_t)(pos - baseFirst)];
            end = B.rows.data() + B.off[(size_t)(pos - baseFirst) + 1];
            return;
        }
        auto it = where.find(ColKey{pos, g, reversed ? 1 : 0});
        if (it == where.end()) { // (not foreseen by the prefetch: asked for by itself)
            run(std::vector<GapAskHost>{GapAskHost{pos, g, reversed ? 1 : 0}});
            it = where.find(ColKey{pos, g, reversed ? 1 : 0});
        }
        const Batch &B = batches[it->second.first];
        begin = B.rows.data() + B.off[it->second.second];
        end = B.rows.data() + B.off[it->second.second + 1];
    }
};

} // namespace

// The ColumnIterator whose sequential state does not fit the device (maxInsertLength > 0, or a visit cache that looks at more
// than the reference genome): api/impl/halColumnIterator.cpp:65-144 (toRight), :246-355 (recursiveUpdate) with :766-819
// (colMapInsert), :357-405 (the indel handlers) over the columns GapColumns delivers.
namespace {
struct ReplayIterator {
    hgx_alignment *al;
    int genome;
    const GenomeTables &G;
    bool unique;
    int64_t maxInsert;
    GapColumns cols;
    int64_t gapChunk, chunkFirst = 0, chunkCount = 0;
    StackEntry base;
    std::vector<StackEntry> upper, insertionStack, deletionStack;
    std::vector<PositionCache> visitCache; // ColumnIterator::VisitCache, by genome (an empty set: the genome has none)
    std::vector<const ColumnRowHost *> column; // the bases of the current column that pass colMapInsert's filters, in its order
    // every base colMapInsert put into the column map since the caller last looked, the ones of abandoned walks included: their
    // sequences stay behind as (empty) keys of the map (resetColMap, :821-825, only empties the sets), and MafBlock::initBlock gives
    // every key an entry
    std::vector<std::pair<int, int64_t>> inserted;
    bool brk = false;
    int64_t leftmostRefPos = 0;
    int refSeqIdx = 0, prevRefSeq = 0;
    int64_t prevRefIndex = 0;

    ReplayIterator(hgx_alignment *a, int g, const ColumnOptions &opt, bool uniq, int64_t maxIns, size_t chunkColumns, ColumnStats *stats)
        : al(a), genome(g), G(a->img.genomes[(size_t)g]), unique(uniq), maxInsert(maxIns), cols(a, g, opt, maxIns, stats),
          gapChunk((int64_t)std::min<size_t>(chunkColumns, (size_t)1 << 17)) { // (every visited base comes back: smaller chunks)
        base.g = g;
        visitCache.resize(a->img.genomes.size());
    }
    // the constructor's / toSite's part (:54-57, :146-165): the stack holds [first, last] of the reference, then the first column
    void start(int64_t first, int64_t last) {
        upper.clear();
        insertionStack.clear();
        deletionStack.clear();
        base.firstIndex = base.index = first;
        base.lastIndex = last;
        base.reversed = false;
        refSeqIdx = G.seqIndexBySite(first);
        toRight();
    }
    StackEntry &top() { return upper.empty() ? base : upper.back(); }
    void nextFreeIndex() { // :749-764
        StackEntry &e = top();
        if (unique || !upper.empty()) {
            const PositionCache &cache = visitCache[(size_t)e.g];
            if (!cache.set.empty())
                while (cache.find(e.index) && e.index <= e.lastIndex)
                    ++e.index;
        }
    }
    void recursiveUpdate() {
        MAF_TICK(4);
        column.clear();
        brk = false;
        leftmostRefPos = base.index;
        const StackEntry e = top();
        if (upper.empty() && (e.index < chunkFirst || e.index >= chunkFirst + chunkCount)) {
            chunkFirst = e.index;
            chunkCount = std::min<int64_t>(gapChunk, base.lastIndex - e.index + 1);
            cols.prefetch(chunkFirst, chunkCount, base.lastIndex);
        }
        const ColumnRowHost *r, *end;
        cols.column(e.g, e.index, e.reversed, r, end);
        unsigned long long open = 0; // levels of the upward chain whose parse-up branch is under way (their deletion check is still to come)
        // After the walk is abandoned (a base seen before): updateNextTopDup's loop over a paralogy cycle does not look at _break
        // (:653-680), so the cycles the abandoned base lies under — inside the subtree of one of their members — still insert
        // their remaining members (colMapInsert and handleInsertion; the members' own subtrees are not walked).  ringAt[d]: the
        // last base at depth d was such a member; after the break ringOn[d]: the cycle at depth d goes on.
        bool ringAt[256], ringOn[256];
        int breakDepth = -1;
        bool memberInserted = false; // the base row before was a member inserted after the break: its insertion event counts
        for (; r < end; ++r) {
            const int kind = r->_pad[0] & 7;
            if (kind == 4)
                continue;
            if (kind == 2 || kind == 3) {
                const int level = r->_pad[1];
                const bool pendingDeletion = kind == 2 && level >= 1 && level < 64 && ((open >> level) & 1ull);
                if (kind == 2 && level >= 1 && level < 64)
                    open &= ~(1ull << level);
                // after the walk was abandoned only the deletion checks of the updateParent calls it was inside are still made
                // (:585-589 is not guarded by _break), and the insertion checks of the cycle members that are still inserted
                if (maxInsert <= 0 || (brk && !pendingDeletion && !(kind == 3 && memberInserted)))
                    continue;
                const int64_t lo = r->pos, hi = r[1].pos;
                if (lo < 0 || hi < lo || hi >= al->img.genomes[(size_t)r->genome].totalLength)
                    continue; // (getInsertedRange's range of a reversed iterator can leave the genome: undefined in the reference, left out)
                if (hi - lo + 1 + e.cumulativeSize <= maxInsert)
                    pushEntry(kind == 2 ? deletionStack : insertionStack, r->genome, lo, hi, r->rev != 0);
                continue;
            }
            const int depth = r->_pad[1];
            const bool ring = (r->_pad[0] & 16) != 0;
            memberInserted = false;
            if (brk) {
                for (int k = depth + 1; k <= breakDepth; ++k)
                    ringOn[k] = false; // (the calls deeper than this base have returned)
                if (!(ring && depth <= breakDepth && ringOn[depth])) {
                    if (depth <= breakDepth)
                        ringOn[depth] = false; // (another call's base at this depth: whatever cycle was there is over)
                    continue;
                }
            } else {
                ringAt[depth] = ring;
            }
            // colMapInsert
            bool updateCache = r->genome == genome;
            if (maxInsert == 0)
                updateCache = updateCache && e.firstIndex < r->pos;
            for (size_t i = 0; i < upper.size() && !updateCache; ++i)
                updateCache = r->genome == upper[i].g;
            if (!unique && maxInsert == 0)
                updateCache = false;
            bool found;
            if (updateCache) {
                found = !visitCache[(size_t)r->genome].insert(r->pos);
            } else {
                const PositionCache &cache = visitCache[(size_t)r->genome];
                found = !cache.set.empty() && cache.find(r->pos);
            }
#ifdef HGX_HOST_PROFILE
            static const bool trace = getenv("HGX_REPLAY_TRACE") != nullptr;
            if (trace)
                fprintf(stderr, "  col g%d idx %lld: row g%d pos %lld rev %d kind %d depth %d ring %d upd %d found %d brk %d\n", e.g, (long long)e.index,
                        r->genome, (long long)r->pos, (int)r->rev, kind, depth, (int)ring, (int)updateCache, (int)found, (int)brk);
#endif
            if (!found && kind == 0) {
                column.push_back(r);
                inserted.emplace_back(r->genome, r->pos);
            }
            if (r->genome == genome)
                leftmostRefPos = std::min(leftmostRefPos, r->pos);
            if (brk) { // a member of a cycle that goes on
                if (found)
                    ringOn[depth] = false; // (:669-672: the loop returns)
                else
                    memberInserted = true;
                continue;
            }
            if (found) {
                brk = true;
                breakDepth = depth;
                for (int k = 0; k < depth; ++k)
                    ringOn[k] = ringAt[k]; // the cycles whose member's subtree this base lies in
                ringOn[depth] = false;     // (a member that is found itself ends its loop)
                continue;
            }
            if ((r->_pad[0] & 8) && depth >= 1 && depth < 64)
                open |= 1ull << depth; // (on the upward chain the depth is the level)
        }
    }
    void toRight() { // :65-144
        prevRefSeq = refSeqIdx;
        prevRefIndex = base.index - G.seqs[(size_t)refSeqIdx].start;
        if (upper.empty() && !top().inBounds())
            return;
        do {
            nextFreeIndex();
            while (!upper.empty() && !top().inBounds()) {
                upper.pop_back();
                nextFreeIndex();
            }
            if (upper.empty() && !top().inBounds())
                return;
            recursiveUpdate();
            StackEntry &e = top();
            e.index += e.reversed ? -1 : 1;
            if (upper.empty()) {
                const SeqInfo &S = G.seqs[(size_t)refSeqIdx];
                if (base.index < S.start || (base.index >= S.start + S.length && base.index < G.totalLength))
                    refSeqIdx = G.seqIndexBySite(base.index);
            }
        } while (brk);
        for (size_t i = deletionStack.size(); i-- > 0;) // pushStackReversed (:121-123)
            upper.push_back(deletionStack[i]);
        deletionStack.clear();
        for (const StackEntry &x : insertionStack)
            upper.push_back(x);
        insertionStack.clear();
        nextFreeIndex();
        while (!upper.empty() && !top().inBounds()) {
            upper.pop_back();
            nextFreeIndex();
        }
    }
    bool lastColumn() const { return upper.empty() && base.index > base.lastIndex; }                                    // :167-169
    bool canonicalOnRef() const { return leftmostRefPos >= base.firstIndex && leftmostRefPos <= base.lastIndex; }       // :210-214
};
} // namespace

void MafExport::convertSequenceGapped(std::ostream &mafStream, hgx_alignment *alignment, int genome, int seq, int64_t first, int64_t last,
                                      const ColumnOptions &opt) {
    (void)seq;
    const auto tStart = std::chrono::steady_clock::now();
    struct Report { // HGX_MAF_TIMING: what the column-by-column path took
        std::chrono::steady_clock::time_point t0;
        const ColumnStats &st;
        int64_t columns;
        ~Report() {
            if (getenv("HGX_MAF_TIMING"))
                std::cerr << "[hgx maf] column by column: " << columns << " reference columns in "
                          << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s (device " << st.rows_ms << " ms, "
                          << st.rows << " rows)" << std::endl;
        }
    } report{tStart, stats, last - first + 1};
    ReplayIterator it(alignment, genome, opt, _unique, _maxRefGap, chunkColumns, &stats);
    if (!_printTree && !getenv("HGX_MAF_MAP_STATE")) {
        // The columns go through the same state machine as the run-compressed export (RunMachine): a column whose bases all continue
        // the column before by one base on their strands — the reference's among them — is held back as one more column of a
        // run, and the run goes through place() when something else comes (the per-column logic inside a run is place()'s:
        // block-length limits, sequence ends).  The keys abandoned walks and unwritten columns leave in the column map go in
        // between two runs, where the reference adds them.
        typedef RunMachine::PRow PRow;
        RunMachine R(*this, mafStream, -1);
        struct Raw {
            int64_t pos;
            int32_t genome;
            bool rev;
        };
        std::vector<Raw> lastRows;       // the bases of the run's last column, in the walk's order
        std::unique_ptr<PRow[]> headRows; // the run's first column, sorted the way the column map holds it
        size_t headCount = 0;
        int64_t runColumns = 0, runRefPos = 0;
        int runRefSeq = -1;
        auto flushRun = [&]() {
            if (runColumns == 0)
                return;
            const PRow *rows = headRows.get();
            R.batch->extra.push_back(std::move(headRows)); // (the log points into it)
            R.refRank = _rank[(size_t)genome][(size_t)runRefSeq];
            int64_t left = runColumns, off = 0;
            for (;;) {
                R.addKeys(rows, headCount);
                const int64_t k = R.place(rows, headCount, left, runRefPos + off);
                left -= k;
                off += k;
                if (left == 0)
                    break;
                rows = R.advance(rows, headCount, k);
            }
            runColumns = 0;
        };
        auto column = [&]() {
            MAF_TICK(5);
            const bool written = !_unique || it.canonicalOnRef();
            const bool otherKeys = it.inserted.size() != it.column.size(); // (bases of abandoned walks, or of a column that is not written)
            bool continues = written && !otherKeys && runColumns > 0 && it.column.size() == lastRows.size() && it.prevRefSeq == runRefSeq &&
                             it.prevRefIndex == runRefPos + runColumns;
            for (size_t i = 0; continues && i < lastRows.size(); ++i) {
                const ColumnRowHost *r = it.column[i];
                continues = r->genome == lastRows[i].genome && (r->rev != 0) == lastRows[i].rev && r->pos == lastRows[i].pos + (lastRows[i].rev ? -1 : 1);
            }
            if (continues) {
                ++runColumns;
                for (Raw &x : lastRows)
                    x.pos += x.rev ? -1 : 1;
                it.inserted.clear();
                return;
            }
            flushRun();
            for (const auto &gp : it.inserted) { // every base colMapInsert put into the column map leaves its key there
                PRow k;
                RunMachine::describe(alignment->img, _rank, k, gp.first, gp.second, false, 0);
                R.addKeys(&k, 1);
            }
            it.inserted.clear();
            if (!written)
                return;
            headCount = it.column.size();
            headRows.reset(new PRow[headCount ? headCount : 1]);
            lastRows.resize(headCount);
            for (size_t i = 0; i < headCount; ++i) {
                const ColumnRowHost *r = it.column[i];
                RunMachine::describe(alignment->img, _rank, headRows[i], r->genome, r->pos, r->rev != 0, (uint32_t)i);
                lastRows[i] = Raw{r->pos, r->genome, r->rev != 0};
            }
            RunMachine::sortColumn(headRows.get(), headCount);
            runColumns = 1;
            runRefPos = it.prevRefIndex;
            runRefSeq = it.prevRefSeq;
        };
        it.start(first, last);
        column();
        while (!it.lastColumn()) {
            it.toRight();
            column();
        }
        flushRun();
        if (R.appendCount > 0)
            R.endBlock();
        R.flush();
        waitPendingWrite();
        mafStream.flush();
#ifdef HGX_HOST_PROFILE
        if (getenv("HGX_MAF_TIMING"))
            std::cerr << "[hgx maf]   Mticks: recursiveUpdate " << g_mafTicks[4] / 1e6 << " (of which prefetch " << g_mafTicks[6] / 1e6 << "), column() "
                      << g_mafTicks[5] / 1e6 << " (initBlock " << g_mafTicks[1] / 1e6 << " canAppend " << g_mafTicks[2] / 1e6 << " place " << g_mafTicks[3] / 1e6
                      << ")" << std::endl;
#endif
        return;
    }
    ColumnMap colMap; // keys persist between columns like ColumnIterator::_colMap (resetColMap only empties the vectors)
    auto fill = [&]() {
        for (auto &kv : colMap)
            kv.second.clear();
        for (const auto &gp : it.inserted)
            colMap[keyOf(gp.first, gp.second)];
        it.inserted.clear();
        for (const ColumnRowHost *r : it.column)
            colMap[keyOf(r->genome, r->pos)].push_back(r);
    };
    auto refKey = [&]() { return Key{_rank[(size_t)genome][(size_t)it.prevRefSeq], genome, it.prevRefSeq}; };
    // MafExport::convertSequence's loop (maf/impl/halMafExport.cpp:51-87)
    size_t appendCount = 0, numBlocks = 0;
    it.start(first, last);
    fill();
    if (!_unique || it.canonicalOnRef()) {
        initBlock(colMap, refKey(), it.prevRefIndex);
        appendColumn(colMap);
        ++appendCount;
    }
    while (!it.lastColumn()) {
        it.toRight();
        fill();
        if (!_unique || it.canonicalOnRef()) {
            if (appendCount == 0)
                initBlock(colMap, refKey(), it.prevRefIndex);
            if (!canAppendColumn(colMap)) {
                if (numBlocks++ % 1000 == 0)
                    for (auto k = colMap.begin(); k != colMap.end();)
                        k = k->second.empty() ? colMap.erase(k) : std::next(k);
                if (appendCount > 0 && (_keepEmptyRefBlocks || !referenceIsAllGaps())) {
                    printBlock(mafStream);
                    mafStream << '\n';
                }
                initBlock(colMap, refKey(), it.prevRefIndex);
            }
            appendColumn(colMap);
            ++appendCount;
        }
    }
    if (appendCount > 0 && (_keepEmptyRefBlocks || !referenceIsAllGaps())) {
        printBlock(mafStream);
        mafStream << std::endl;
    }
}

// hal2maf --global: MafExport::convertEntireAlignment (maf/impl/halMafExport.cpp:90-153) — every column of the alignment once: the
// leaves in breadth-first order (getLeafGenomes, api/impl/halCommon.cpp:197-207), each walked over its whole genome by a --unique
// iterator that is handed the visit cache of the leaves before it; a column with a base of an earlier leaf was written then.
void MafExport::convertEntireAlignment(std::ostream &mafStream, hgx_alignment *alignment) {
    if (_al != alignment) {
        _al = alignment;
        buildRanks();
    }
    const Image &img = alignment->img;
    if (mafStream.tellp() <= std::streampos(0)) // writeHeader (:15-23), unconditionally here (:97)
        mafStream << "##maf version=1 scoring=N/A\n"
                  << "# hal " << img.newick << std::endl
                  << std::endl;
    std::vector<int> leaves; // Alignment::getLeafNamesBelow(root): breadth first, children in their order
    {
        std::deque<int> queue(1, img.root());
        while (!queue.empty()) {
            const int g = queue.front();
            queue.pop_front();
            if (img.genomes[(size_t)g].children.empty() && g != img.root())
                leaves.push_back(g);
            for (int c : img.genomes[(size_t)g].children)
                queue.push_back(c);
        }
    }
    ColumnOptions opt;
    opt.noDupes = _noDupes;
    opt.noAncestors = _noAncestors;
    opt.onlyOrthologs = _onlyOrthologs;
    std::vector<PositionCache> visitCache(img.genomes.size());
    size_t appendCount = 0, numBlocks = 0;
    for (int genome : leaves) {
        const GenomeTables &G = img.genomes[(size_t)genome];
        if (G.totalLength == 0)
            continue;
        ReplayIterator it(alignment, genome, opt, true, 0, chunkColumns, &stats);
        // every leaf has an iterator of its own (:106-110), hence a column map of its own; the first column its constructor walks
        // (with the iterator's own, empty cache, which setVisitCache then replaces) leaves its sequences behind as empty keys of that
        // map (toSite's defragment, :160, sees them filled), and initBlock gives every empty key an entry (halMafBlock.cpp:311-322)
        ColumnMap colMap;
        it.start(0, G.totalLength - 1);
        it.visitCache.swap(visitCache);
        it.start(0, G.totalLength - 1); // toSite(0, length - 1), :111-113
        auto fill = [&]() {
            for (auto &kv : colMap)
                kv.second.clear();
            for (const auto &gp : it.inserted) // (the constructor's column among them)
                colMap[keyOf(gp.first, gp.second)];
            it.inserted.clear();
            for (const ColumnRowHost *r : it.column)
                colMap[keyOf(r->genome, r->pos)].push_back(r);
        };
        auto refKey = [&]() { return Key{_rank[(size_t)genome][(size_t)it.prevRefSeq], genome, it.prevRefSeq}; };
        for (;;) {
            fill();
            if (appendCount == 0)
                initBlock(colMap, refKey(), it.prevRefIndex);
            if (!canAppendColumn(colMap)) {
                if (numBlocks++ % 1000 == 0)
                    for (auto k = colMap.begin(); k != colMap.end();)
                        k = k->second.empty() ? colMap.erase(k) : std::next(k);
                if (appendCount > 0) {
                    printBlock(mafStream);
                    mafStream << '\n';
                }
                initBlock(colMap, refKey(), it.prevRefIndex);
            }
            appendColumn(colMap);
            ++appendCount;
            if (it.lastColumn())
                break;
            it.toRight();
        }
        visitCache.swap(it.visitCache);
    }
    if (appendCount > 0) {
        printBlock(mafStream);
        mafStream << std::endl;
    }
}

// hal2mafMP.py (maf/hal2mafMP.py:63-79 computeSlices, :176-190 concatenateSlices) with devices in the place of processes
void mafExportSliced(std::ostream &os, const std::vector<hgx_alignment *> &handles, int genome, int sequence, int64_t start, int64_t length,
                     int64_t sliceSize, const MafExportSettings &cfg, const std::set<int> &targets) {
    if (handles.empty())
        throw std::runtime_error("mafExportSliced: no handles");
    const GenomeTables &G = handles[0]->img.genomes[(size_t)genome];
    struct Slice {
        int seq;
        int64_t start, length;
        std::string text, error;
    };
    std::vector<Slice> slices;
    for (int s = 0; s < (int)G.seqs.size(); ++s) {
        if (sequence >= 0 && s != sequence)
            continue;
        const int64_t seqLen = G.seqs[(size_t)s].length;
        if (seqLen == 0 && sequence < 0)
            continue; // (hal2maf.cpp:200-205 skips nothing, but a zero-length sequence cannot be converted: halMafExport.cpp:35-37)
        if (start >= seqLen || start + length > seqLen)
            throw std::runtime_error("Invalid range specified for convertGenome");
        const int64_t inLength = length < 1 ? seqLen - start : length;
        int64_t size = sliceSize > 0 ? sliceSize : (inLength + (int64_t)handles.size() - 1) / (int64_t)handles.size();
        if (size < 1 || size >= inLength) {
            slices.push_back(Slice{s, start, inLength, "", ""});
            continue;
        }
        for (int64_t i = 0; i < inLength / size; ++i)
            slices.push_back(Slice{s, start + i * size, size, "", ""});
        if (inLength % size > 0)
            slices.push_back(Slice{s, start + (inLength / size) * size, inLength % size, "", ""});
    }
    int64_t allColumns = 0;
    for (const Slice &sl : slices)
        allColumns += sl.length;
    // A slice's text grows in memory of the text allocator (huge pages, mremap: hgx_textmem.hpp) where the rendering can be given room
    // for a batch at once (BulkSink); it used to grow in a std::ostringstream, be copied out of it, and be copied once more line by line
    // to drop the header — three passes over seventeen megabytes a slice, more than the export itself took.
    struct SliceText : std::streambuf, BulkSink {
        char *p = nullptr;
        size_t n = 0, cap = 0;
        ~SliceText() override { textFree(p); }
        bool reserve(size_t need) {
            if (need <= cap)
                return true;
            size_t c = cap ? cap : (size_t)1 << 20;
            while (c < need)
                c += c < ((size_t)1 << 28) ? c : ((size_t)1 << 28);
            char *q = static_cast<char *>(textRealloc(p, c));
            if (!q)
                return false;
            p = q;
            cap = c;
            return true;
        }
        char *room(size_t k) override {
            if (!reserve(n + k))
                return nullptr;
            char *q = p + n;
            n += k;
            return q;
        }
        std::streamsize xsputn(const char *src, std::streamsize k) override {
            if (k <= 0)
                return 0;
            char *q = room((size_t)k);
            if (!q)
                return 0;
            memcpy(q, src, (size_t)k);
            return k;
        }
        int_type overflow(int_type c) override {
            if (traits_type::eq_int_type(c, traits_type::eof()))
                return traits_type::not_eof(c);
            const char ch = traits_type::to_char_type(c);
            return xsputn(&ch, 1) == 1 ? c : traits_type::eof();
        }
    };
    std::vector<std::unique_ptr<SliceText>> texts(slices.size());
    std::vector<size_t> skip(slices.size(), 0); // concatenateSlices (hal2mafMP.py:176-190): of every slice but the first the lines that start with '#' are dropped
    std::atomic<size_t> next{0};
    auto work = [&](hgx_alignment *h) {
        for (size_t i; (i = next.fetch_add(1)) < slices.size();) {
            Slice &sl = slices[i];
            try {
                MafExport me;
                me.setNoDupes(cfg.noDupes);
                me.setNoAncestors(cfg.noAncestors);
                me.setUcscNames(cfg.ucscNames);
                me.setOnlyOrthologs(cfg.onlyOrthologs);
                me.setKeepEmptyRefBlocks(cfg.keepEmptyRefBlocks);
                me.setUnique(cfg.unique);
                me.setMaxBlockLength(cfg.maxBlockLength);
                me.setMaxRefGap(cfg.maxRefGap);
                me.setPrintTree(cfg.printTree);
                me.setExportHint(allColumns / (int64_t)handles.size());
                texts[i].reset(new SliceText);
                std::ostream text(texts[i].get());
                me.convertSequence(text, h, genome, sl.seq, sl.start, sl.length, targets);
                text.flush();
                if (!text.good())
                    throw std::runtime_error("out of memory for a slice's text");
                if (i != 0) {
                    // (a MAF text's '#' lines are its header: every other line begins with 'a', 's' or is empty — halMafBlock.cpp:499-520 —
                    // so the lines to drop are the ones the text begins with)
                    const char *p = texts[i]->p;
                    const size_t n = texts[i]->n;
                    size_t a = 0;
                    while (a < n && p[a] == '#') {
                        const void *nl = memchr(p + a, '\n', n - a);
                        a = nl ? (size_t)((const char *)nl - p) + 1 : n;
                    }
                    skip[i] = a;
                }
            } catch (std::exception &e) {
                sl.error = e.what();
            }
        }
    };
    // A slice is an export of a batch or two: its device stage, its walk and its rendering follow one another with nothing to run
    // beside them.  A few slices at a time a handle fill each other's waits (HGX_MAF_MULTI_PER_HANDLE; the engine's shared state — the
    // handle's tracks, the caches of device and page-locked memory, the rendering streams — is behind mutexes, a slice's buffers are its own).
    size_t perHandle = 3; // (55 slices of config 3 over two handles on the box: 0.42 s one at a time, 0.30 two, 0.26 three)
    if (const char *e = getenv("HGX_MAF_MULTI_PER_HANDLE"))
        perHandle = (size_t)std::max(1, atoi(e));
    // ... and at most four slices at a time a DEVICE (HGX_MAF_MULTI_PER_DEVICE): beyond that the slices' device stages and renderings
    // only queue behind each other (0.25-0.27 s for config 3's 55 slices with four as with six).  (Round 6, the last day: about every
    // third run of config 3's leg was lost in here — to textRealloc's race, hgx_textmem.cpp, and to device blocks released behind a
    // launch in columnsHeadRowsSweep, hgx_columns.hip: profiles/r06_notes.md 12)
    size_t perDevice = 4;
    if (const char *e = getenv("HGX_MAF_MULTI_PER_DEVICE"))
        perDevice = (size_t)std::max(1, atoi(e));
    std::map<int, size_t> handlesOn, startedOn;
    for (hgx_alignment *h : handles)
        ++handlesOn[h->dev ? h->dev->device : -1];
    std::vector<std::thread> pool;
    startedOn[handles[0]->dev ? handles[0]->dev->device : -1] = 1; // (this thread: handles[0]'s first)
    for (size_t d = 0; d < handles.size(); ++d) {
        const int dev = handles[d]->dev ? handles[d]->dev->device : -1;
        const size_t share = std::max<size_t>(1, perDevice / handlesOn[dev]); // (a handle's part of its device's four)
        for (size_t k = 0; k < std::min(perHandle, share); ++k)
            if (d + k > 0 && pool.size() + 1 < slices.size() && startedOn[dev] < perDevice) {
                pool.emplace_back(work, handles[d]);
                ++startedOn[dev];
            }
    }
    work(handles[0]);
    for (std::thread &t : pool)
        t.join();
    for (const Slice &sl : slices)
        if (!sl.error.empty())
            throw std::runtime_error(sl.error);
    // the slices' texts into the output: side by side where it gives room for all of them at once
    size_t total = 0;
    std::vector<size_t> at(slices.size(), 0);
    for (size_t i = 0; i < slices.size(); ++i) {
        at[i] = total;
        total += texts[i] ? texts[i]->n - skip[i] : 0;
    }
    BulkSink *const sink = dynamic_cast<BulkSink *>(os.rdbuf());
    char *dst = sink && total >= ((size_t)1 << 20) ? sink->room(total) : nullptr;
    if (!dst) {
        for (size_t i = 0; i < slices.size(); ++i)
            if (texts[i])
                os.write(texts[i]->p + skip[i], (std::streamsize)(texts[i]->n - skip[i]));
        return;
    }
    std::atomic<size_t> nextCopy{0};
    auto copy = [&]() {
        for (size_t i; (i = nextCopy.fetch_add(1)) < slices.size();)
            if (texts[i]) {
                memcpy(dst + at[i], texts[i]->p + skip[i], texts[i]->n - skip[i]);
                texts[i].reset(); // (its pages go back while the others are copied)
            }
    };
    std::vector<std::thread> copiers;
    for (unsigned t = 1; t < std::min<unsigned>(8u, std::max(1u, hostThreads())) && t < slices.size(); ++t)
        copiers.emplace_back(copy);
    copy();
    for (std::thread &t : copiers)
        t.join();
}

// maf/impl/halMafBed.cpp:24-52 driven by BedScanner::scan (liftover/impl/halBedScanner.cpp:40-61)
void MafExport::convertBed(std::ostream &mafStream, hgx_alignment *alignment, int genome, std::istream &in, const std::set<int> &targets) {
    const GenomeTables &G = alignment->img.genomes[(size_t)genome];
    auto skipWhiteSpaces = [](std::istream &s) {
        while (s.good() && std::isspace((char)s.peek()))
            s.get();
    };
    BedLine bedLine;
    std::string lineBuffer;
    size_t lineNumber = 0;
    skipWhiteSpaces(in);
    while (in.good()) {
        ++lineNumber;
        try {
            std::getline(in, lineBuffer);
            bedLine.parse(lineBuffer, 0);
        } catch (std::runtime_error &e) {
            throw std::runtime_error(std::string(e.what()) + " in input bed line " + std::to_string(lineNumber));
        }
        const int seq = G.seqIndexByName(bedLine.chrName);
        if (seq < 0) {
            std::cerr << "Line " << lineNumber << ": BED sequence " << bedLine.chrName << " not found in genome " << G.name << '\n';
        } else if (bedLine.bedType <= 9) {
            if (bedLine.end <= bedLine.start || bedLine.end > G.seqs[(size_t)seq].length)
                std::cerr << "Line " << lineNumber << ": BED coordinates invalid\n";
            else
                convertSequence(mafStream, alignment, genome, seq, bedLine.start, bedLine.end - bedLine.start, targets);
        } else {
            for (size_t i = 0; i < bedLine.blocks.size(); ++i) {
                const BedBlock &b = bedLine.blocks[i];
                if (b.length == 0 || bedLine.start + b.start + b.length >= G.seqs[(size_t)seq].length)
                    std::cerr << "Line " << lineNumber << ", block " << i << ": BED coordinates invalid\n";
                else
                    convertSequence(mafStream, alignment, genome, seq, bedLine.start + b.start, b.length, targets);
            }
        }
        skipWhiteSpaces(in);
    }
}

} // namespace hgx
