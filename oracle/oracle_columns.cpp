// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).
#include "oracle_columns.hpp"
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <chrono>

namespace orc {

static const char dnaUnpackMap[16] = {'a', 'c', 'g', 't', 'n', '\0', '\0', '\0', 'A', 'C', 'G', 'T', 'N', '\0', '\0', '\0'};

// api/inc/halCommon.h:45-75
static char reverseComplement(char c) {
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
        break;
    }
    return c;
}

char dnaBase(const Alignment &al, const Dna &d) {
    const std::vector<u8> &p = al.genomes[(size_t)d.g].dna;
    if (p.empty()) // an HGX image written without DNA (not a reference format): every base reads as N, as in the product
        return 'N';
    u8 b = p[(size_t)(d.pos >> 1)];
    char c = dnaUnpackMap[(d.pos & 1) ? (b & 0x0F) : (b >> 4)]; // halCommon.h:187-190
    return d.rev ? reverseComplement(c) : c;
}

static int seqIndexBySite(const Genome &G, i64 pos) {
    const Sequence *s = G.seqBySite(pos);
    return s ? (int)(s - G.seqs.data()) : -1;
}

// halColumnIterator.cpp:18-57
ColumnIterator::ColumnIterator(const Alignment *a, int reference, const std::set<int> *tgts, i64 columnIndex, i64 lastColumnIndex,
                               bool noDupes_, bool noAncestors_, bool onlyOrthologs_, bool unique_, i64 maxInsertLength_)
    : al(a), refGenome(reference), noDupes(noDupes_), noAncestors(noAncestors_), onlyOrthologs(onlyOrthologs_), unique(unique_),
      maxInsertLength(maxInsertLength_) {
    base.g = reference;
    seqIdx = seqIndexBySite(al->genomes[(size_t)reference], columnIndex);
    refSeqIdx = seqIdx;
    if (tgts != nullptr && !tgts->empty()) {
        targets = *tgts;
        targets.insert(reference);
        getGenomesInSpanningTree(*al, targets, scope);
    }
    firstIndex = index = columnIndex;
    lastIndex = lastColumnIndex;
    prevRefSeq = seqIdx;
    prevRefIndex = 0;
    toRight();
}

// ColumnIteratorStack::push (halColumnIteratorStack.h:112-121) on one of the indel stacks
void ColumnIterator::pushEntry(std::vector<Entry> &st, int g, int seqIdx, i64 index, i64 lastIndex, bool reversed) {
    Entry e;
    e.g = g;
    e.seqIdx = seqIdx;
    e.cumulativeSize = st.empty() ? 0 : st.back().cumulativeSize + lastIndex - index + 1;
    e.firstIndex = index;
    e.index = reversed ? lastIndex : index;
    e.lastIndex = lastIndex;
    e.reversed = reversed;
    st.push_back(e);
}

// halColumnIterator.cpp:749-764
void ColumnIterator::nextFreeIndex() {
    Entry &e = top();
    i64 idx = e.index;
    if (unique || !upper.empty()) {
        auto it = visitCache.find(e.g);
        if (it != visitCache.end()) {
            bool found = it->second.find(idx);
            while (found && idx <= e.lastIndex) {
                ++idx;
                found = it->second.find(idx);
            }
        }
    }
    e.index = idx;
}

// halColumnIterator.cpp:65-144
void ColumnIterator::toRight() {
    const Genome &G = al->genomes[(size_t)refGenome];
    prevRefSeq = refSeqIdx;
    prevRefIndex = index - G.seqs[(size_t)refSeqIdx].start;
    if (upper.empty() && !topInBounds())
        return;
    do {
        nextFreeIndex();
        while (!upper.empty() && !topInBounds()) {
            upper.pop_back();
            nextFreeIndex();
        }
        if (upper.empty() && !topInBounds())
            return;
        recursiveUpdate();
        Entry &e = top();
        if (e.reversed)
            e.index--;
        else
            e.index++;
        if (upper.empty()) { // jump to the next sequence of the genome if necessary (:109-118)
            const Sequence &seq = G.seqs[(size_t)seqIdx];
            if (index < seq.start || (index >= seq.start + seq.length && index < G.totalLength)) {
                seqIdx = seqIndexBySite(G, index);
                refSeqIdx = seqIdx;
            }
        }
    } while (brk);
    // push the indel stacks (:121-123)
    for (size_t i = deletionStack.size(); i-- > 0;)
        upper.push_back(deletionStack[i]);
    deletionStack.clear();
    for (size_t i = 0; i < insertionStack.size(); ++i)
        upper.push_back(insertionStack[i]);
    insertionStack.clear();
    nextFreeIndex();
    while (!upper.empty() && !topInBounds()) {
        upper.pop_back();
        nextFreeIndex();
    }
}

// halColumnIterator.cpp:193-208
void ColumnIterator::defragment() {
    for (auto i = colMap.begin(); i != colMap.end();) {
        if (i->second.empty())
            i = colMap.erase(i);
        else
            ++i;
    }
}

// halColumnIterator.cpp:766-819
bool ColumnIterator::colMapInsert(const SegIt &it) {
    const int g = it.g;
    const i64 pos = it.getStartPosition();
    static const bool trace = getenv("ORC_TRACE") != nullptr; // (debugging aid: the order bases reach the column map in)
    if (trace)
        fprintf(stderr, "  col g%d idx %lld: insert g%d pos %lld rev %d\n", refGenome, (long long)top().index, g, (long long)pos, (int)it.rev);
    bool updateCache = g == refGenome; // all reference bases are added to the cache ...
    if (maxInsertLength == 0)          // ... unless indels are not done: then only the ones right of the starting point
        updateCache = updateCache && top().firstIndex < pos;
    for (size_t i = 0; i < upper.size() && !updateCache; ++i)
        if (g == upper[i].g)
            updateCache = true;
    if (!unique && maxInsertLength == 0)
        updateCache = false;
    bool found = false;
    auto cacheIt = visitCache.find(g);
    if (updateCache) {
        if (cacheIt == visitCache.end())
            cacheIt = visitCache.emplace(g, PositionCache()).first;
        found = !cacheIt->second.insert(pos);
    } else {
        found = cacheIt != visitCache.end() && cacheIt->second.find(pos);
    }
    if (trace)
        fprintf(stderr, "      upd %d found %d cache-has-genome %d\n", (int)updateCache, (int)found, (int)(visitCache.find(g) != visitCache.end()));
    if (!found && (!noAncestors || al->genomes[(size_t)g].children.empty()) && (targets.empty() || targets.count(g))) {
        SeqKey k{al, g, seqIndexBySite(al->genomes[(size_t)g], pos)};
        colMap[k].push_back(Dna{g, pos, it.rev});
    }
    if (g == refGenome)
        leftmostRefPos = std::min(leftmostRefPos, pos);
    return !found;
}

// halColumnIterator.cpp:246-355
void ColumnIterator::recursiveUpdate() {
    for (auto &kv : colMap) // resetColMap :821-825 (keys persist)
        kv.second.clear();
    brk = false;
    leftmostRefPos = index;
    const Entry &e = top();
    const Genome &G = al->genomes[(size_t)e.g];
    const Sequence &refSeq = G.seqs[(size_t)e.seqIdx];
    SegIt it;
    it.al = al;
    it.g = e.g;
    if (refSeq.numTop > 0) {
        it.top = true;
        it.toSite(e.index, true);
        if (e.reversed)
            it.toReverseInPlace();
        if (!colMapInsert(it)) {
            brk = true;
            return;
        }
        handleDeletion(it);
        updateParent(it);
        if (!onlyOrthologs)
            updateNextTopDup(it);
        updateParseDown(it);
    } else {
        it.top = false;
        it.toSite(e.index, true);
        if (e.reversed)
            it.toReverseInPlace();
        if (!colMapInsert(it)) {
            brk = true;
            return;
        }
        for (size_t child = 0; child < G.children.size(); ++child)
            updateChild(it, (i64)child);
    }
}

// ---- Segment::isFirst / isLast (api/mmap_impl/mmapTopSegment.h:102-111, mmapBottomSegment.h:111-120) through an iterator
// (halSegmentIterator.cpp:110-116) ----
static bool segIsFirst(const SegIt &s) {
    const Sequence *q = s.getSequence();
    return s.idx == 0 || s.idx == (s.top ? q->topStart : q->botStart);
}
static bool segIsLast(const SegIt &s) {
    const Sequence *q = s.getSequence();
    return s.idx == s.numSegs() || s.idx == (s.top ? q->topStart + q->numTop : q->botStart + q->numBot) - 1;
}
static bool itIsFirst(const SegIt &s) {
    return !s.rev ? segIsFirst(s) : segIsLast(s);
}
static bool itIsLast(const SegIt &s) {
    return !s.rev ? segIsLast(s) : segIsFirst(s);
}

// Rearrangement (api/impl/halRearrangement.cpp) as the column iterator holds it: getRearrangement(0, 0, 1., true)
// (halColumnIterator.cpp:34-39) = gap threshold 0, atomic — every gapped iterator is one whole segment, the "next ungapped"
// moves of halGappedTop/BottomSegmentIterator.cpp never step (no segment is shorter than one base) and extendLeft / extendRight
// return at once.  What is left of the gapped iterators: a SegIt over a whole segment.
namespace {
struct AtomicRearrangement {
    SegIt cur, next, left, right, leftParent, rightParent, curParent;
    // GappedBottomSegmentIterator::adjacentTo (halGappedBottomSegmentIterator.cpp:293-337), one segment each
    static bool adjacentTo(const SegIt &self, const SegIt &other) {
        for (int pass = 0; pass < 2; ++pass) {
            const SegIt t = self; // _left, then _right: the same segment
            if (pass == 0 ? !itIsFirst(t) : !itIsLast(t)) {
                SegIt t2 = other;
                if (!itIsFirst(t2)) {
                    t2.toLeft();
                    if (t.idx == t2.idx)
                        return true;
                }
                t2 = other;
                if (!itIsLast(t2)) {
                    t2.toRight();
                    if (t.idx == t2.idx)
                        return true;
                }
            }
        }
        return false;
    }
    void resetStatus(const SegIt &topSegment) { // :243-270
        cur = next = left = right = topSegment;
    }
    // :458-516; true: leftParent holds the deletion candidate
    bool scanDeletionCycle(const SegIt &topSegment) {
        resetStatus(topSegment);
        const bool first = itIsFirst(cur), last = itIsLast(cur);
        if (!cur.hasParent() || (first && last))
            return false;
        if (last) {
            leftParent.toParent(cur);
            if (!itIsFirst(leftParent)) {
                leftParent.toLeft();
                return true;
            }
            if (!itIsLast(leftParent)) {
                leftParent.toRight();
                return true;
            }
        } else {
            leftParent.toParent(cur);
            right.toRight();
            if (!right.hasParent())
                return false;
            rightParent.toParent(right);
            if (leftParent.getSequence() == rightParent.getSequence()) {
                if (leftParent.rev)
                    leftParent.toReverse();
                if (rightParent.rev)
                    rightParent.toReverse();
                if (rightParent.idx < leftParent.idx)
                    std::swap(leftParent, rightParent);
                if (itIsLast(leftParent))
                    return false;
                leftParent.toRight();
                return adjacentTo(leftParent, rightParent);
            }
        }
        return false;
    }
    // :133-140
    bool identifyDeletionFromLeftBreakpoint(const SegIt &topSegment) {
        if (!scanDeletionCycle(topSegment))
            return false;
        const i64 slot = leftParent.G().childSlotOf(topSegment.g); // the gapped bottom iterator's _childIndex (toParent)
        return !leftParent.hasChild(slot);
    }
    std::pair<i64, i64> getDeletedRange() const { // :142-154 (left == right: one segment)
        if (!leftParent.rev)
            return std::make_pair(leftParent.getStartPosition(), leftParent.getEndPosition());
        return std::make_pair(leftParent.getEndPosition(), leftParent.getStartPosition());
    }
    // :386-456; true: cur holds the insertion candidate
    bool scanInsertionCycle(const SegIt &topSegment) {
        resetStatus(topSegment);
        while (!next.hasParent() && !itIsLast(next)) {
            right = next;
            right.toRight();
            if (!right.hasParent())
                next = right;
            else
                break;
        }
        right = next;
        const bool first = itIsFirst(cur), last = itIsLast(right);
        if (first && last)
            return false;
        if (first) {
            right.toRight();
            if (!cur.hasParent()) {
                return true;
            } else if (right.hasParent()) {
                curParent.toParent(cur);
                rightParent.toParent(right);
                return !adjacentTo(rightParent, curParent);
            }
        } else if (last) {
            left.toLeft();
            if (!cur.hasParent()) {
                return true;
            } else if (left.hasParent()) {
                curParent.toParent(cur);
                leftParent.toParent(left);
                return !adjacentTo(leftParent, curParent);
            }
        } else {
            left.toLeft();
            right.toRight();
            if (left.hasParent() && right.hasParent()) {
                leftParent.toParent(left);
                rightParent.toParent(right);
                if (adjacentTo(leftParent, rightParent))
                    return true;
                else if (itIsFirst(leftParent) || itIsLast(leftParent))
                    return leftParent.getSequence() == rightParent.getSequence();
                else if (itIsFirst(rightParent) || itIsLast(rightParent))
                    return leftParent.getSequence() == rightParent.getSequence();
            }
        }
        return false;
    }
    // :156-163
    bool identifyInsertionFromLeftBreakpoint(const SegIt &topSegment) {
        return scanInsertionCycle(topSegment) && !cur.hasParent();
    }
    std::pair<i64, i64> getInsertedRange() const { // :165-176, as written (left == right; a reversed iterator's "start" is its high end)
        std::pair<i64, i64> range;
        range.first = cur.getStartPosition();
        range.second = cur.getStartPosition() + (cur.getLength() - 1);
        if (range.first >= range.second)
            std::swap(range.first, range.second);
        return range;
    }
};
} // namespace

// halColumnIterator.cpp:357-382
bool ColumnIterator::handleDeletion(const SegIt &inputTopSegIt) {
    if (maxInsertLength > 0 && inputTopSegIt.hasParent()) {
        SegIt topIt = inputTopSegIt;
        if (topIt.eo == 0) { // only immediately left of the breakpoint
            topIt.slice(0, 0);
            AtomicRearrangement rea;
            if (rea.identifyDeletionFromLeftBreakpoint(topIt) &&
                rea.leftParent.getLength() + top().cumulativeSize <= maxInsertLength) {
                const std::pair<i64, i64> deletedRange = rea.getDeletedRange();
                SegIt botSegIt;
                botSegIt.toParent(inputTopSegIt);
                const Genome &P = botSegIt.G();
                if (deletedRange.first < 0 || deletedRange.second >= P.totalLength)
                    return false;
                pushEntry(deletionStack, botSegIt.g, seqIndexBySite(P, botSegIt.segStart()), deletedRange.first, deletedRange.second, botSegIt.rev);
                return true;
            }
        }
    }
    return false;
}

// halColumnIterator.cpp:384-405
bool ColumnIterator::handleInsertion(const SegIt &inputTopSegIt) {
    if (maxInsertLength > 0 && inputTopSegIt.hasParent()) {
        SegIt topIt = inputTopSegIt;
        const bool reversed = topIt.rev;
        if (topIt.eo == 0 && !itIsLast(topIt)) { // only immediately left of the break
            topIt.slice(0, 0);
            topIt.toRight();
            AtomicRearrangement rea;
            if (rea.identifyInsertionFromLeftBreakpoint(topIt) && rea.cur.getLength() + top().cumulativeSize <= maxInsertLength) {
                const std::pair<i64, i64> insertedRange = rea.getInsertedRange();
                const Genome &G = topIt.G();
                // (the reference computes a reversed iterator's range from its high end upwards, :165-176; a range that leaves
                // the genome is undefined behaviour there — a column iterator over positions that do not exist — and is left
                // out here, as in the product)
                if (insertedRange.first < 0 || insertedRange.second >= G.totalLength)
                    return false;
                pushEntry(insertionStack, topIt.g, seqIndexBySite(G, topIt.segStart()), insertedRange.first, insertedRange.second, reversed);
            }
        }
    }
    return false;
}

// halColumnIterator.cpp:556-605
void ColumnIterator::updateParent(const SegIt &top) {
    const int genome = top.g;
    if (!brk && top.hasParent() && parentInScope(genome) && (!noDupes || top.isCanonicalParalog())) {
        SegIt parent;
        parent.toParent(top);
        if (!colMapInsert(parent)) {
            brk = true;
            return;
        }
        updateParseUp(parent);
        if (parent.G().bTopParse[(size_t)parent.idx] != NULL_INDEX) { // hasParseUp and its linked top iterator exists (:587-589)
            SegIt topParse;
            topParse.toParseUp(parent);
            handleDeletion(topParse);
        }
        const Genome &P = al->genomes[(size_t)parent.g];
        for (size_t i = 0; i < P.children.size(); ++i) {
            if (P.children[i] != genome)
                updateChild(parent, (i64)i);
        }
    }
}

// halColumnIterator.cpp:607-640
void ColumnIterator::updateChild(const SegIt &bot, i64 slot) {
    if (!brk && bot.hasChild(slot) && childInScope(bot.g, slot)) {
        SegIt child;
        child.toChild(bot, slot);
        if (!colMapInsert(child)) {
            brk = true;
            return;
        }
        handleInsertion(child);
        updateNextTopDup(child);
        updateParseDown(child);
    }
}

// halColumnIterator.cpp:642-681
void ColumnIterator::updateNextTopDup(const SegIt &top) {
    const Genome &G = top.G();
    if (brk || noDupes || G.tParalogy[(size_t)top.idx] == NULL_INDEX || G.parent < 0 || !parentInScope(top.g))
        return;
    const i64 firstIndexSeg = top.idx;
    SegIt cur = top;
    do {
        SegIt dup = cur;
        dup.toNextParalogy();
        if (!colMapInsert(dup)) {
            brk = true;
            return;
        }
        handleInsertion(dup);
        updateParseDown(dup);
        cur = dup;
    } while (G.tParalogy[(size_t)cur.idx] != NULL_INDEX && G.tParalogy[(size_t)cur.idx] != firstIndexSeg);
}

// halColumnIterator.cpp:683-709
void ColumnIterator::updateParseUp(const SegIt &bot) {
    if (!brk && bot.G().bTopParse[(size_t)bot.idx] != NULL_INDEX) { // hasParseUp
        SegIt top;
        top.toParseUp(bot);
        updateParent(top);
        if (!onlyOrthologs)
            updateNextTopDup(top);
    }
}

// halColumnIterator.cpp:711-744
void ColumnIterator::updateParseDown(const SegIt &top) {
    if (!brk && top.G().tBotParse[(size_t)top.idx] != NULL_INDEX) { // hasParseDown
        SegIt bot;
        bot.toParseDown(top);
        const Genome &G = bot.G();
        for (size_t i = 0; i < G.children.size(); ++i)
            updateChild(bot, (i64)i);
    }
}

// ---------------------------------------------------------------------------------------------
// alignmentDepth/halAlignmentDepth.cpp:215-308
static void printDepthSequence(std::ostream &os, const Alignment &al, int genome, int seqIdx, const std::set<int> &targetSet,
                               i64 start, i64 length, i64 step, bool countDupes, bool noAncestors) {
    const Sequence &sequence = al.genomes[(size_t)genome].seqs[(size_t)seqIdx];
    i64 seqLen = sequence.length;
    if (seqLen == 0)
        return;
    if (length == 0)
        length = seqLen - start;
    i64 last = start + length;
    if (last > seqLen)
        throw std::runtime_error("Specified range is out of range for sequence " + sequence.name);
    i64 pos = start;
    ColumnIterator colIt(&al, genome, &targetSet, pos + sequence.start, last - 1 + sequence.start, false, noAncestors, false);
    os << "fixedStep chrom=" << sequence.name << " start=" << start + 1 << " step=" << step << "\n";
    pos += sequence.start;
    last += sequence.start;
    std::set<int> genomeSet;
    while (pos <= last) {
        genomeSet.clear();
        i64 count = 0;
        for (auto &kv : colIt.colMap) {
            if (countDupes)
                count += (i64)kv.second.size();
            else if (!kv.second.empty())
                genomeSet.insert(kv.first.g);
        }
        if (!countDupes)
            count = (i64)genomeSet.size();
        --count;
        os << count << '\n';
        if (colIt.lastColumn())
            break;
        pos += step;
        if (step == 1) {
            colIt.toRight();
            if (pos % 1000 == 0)
                colIt.defragment();
        } else {
            // ColumnIterator::toSite(pos, last) (halColumnIterator.cpp:146-165): restart at a non-contiguous site
            if (pos > last || pos >= al.genomes[(size_t)genome].totalLength) // past the range: the loop test ends it
                break;
            colIt.seqIdx = seqIndexBySite(al.genomes[(size_t)genome], pos);
            colIt.refSeqIdx = colIt.seqIdx;
            colIt.defragment();
            colIt.firstIndex = colIt.index = pos;
            colIt.lastIndex = last;
            colIt.toRight();
        }
    }
}

// alignmentDepth/halAlignmentDepth.cpp:318-347
void printDepthGenome(std::ostream &os, const Alignment &al, int genome, int sequence, const std::set<int> &targetSet, i64 start,
                      i64 length, i64 step, bool countDupes, bool noAncestors) {
    const Genome &G = al.genomes[(size_t)genome];
    if (sequence >= 0) {
        printDepthSequence(os, al, genome, sequence, targetSet, start, length, step, countDupes, noAncestors);
        return;
    }
    if (start + length > G.totalLength)
        throw std::runtime_error("Specified range is out of range for genome " + G.name);
    if (length == 0)
        length = G.totalLength - start;
    i64 runningLength = 0;
    for (size_t s = 0; s < G.seqs.size(); ++s) {
        const Sequence &seq = G.seqs[s];
        i64 seqLen = seq.length, seqStart = seq.start;
        if (start + length >= seqStart && start < seqStart + seqLen && runningLength < length) {
            i64 readStart = seqStart >= start ? 0 : start - seqStart;
            i64 readLen = std::min(seqLen - readStart, length);
            readLen = std::min(readLen, length - runningLength);
            printDepthSequence(os, al, genome, (int)s, targetSet, readStart, readLen, step, countDupes, noAncestors);
            runningLength += readLen;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MAF
MafExport::~MafExport() {
    delete tree;
    for (auto &kv : entries)
        delete kv.second;
}

std::string MafExport::getName(const SeqKey &k) const { // halMafBlock.h:128-130
    const Genome &G = k.al->genomes[(size_t)k.g];
    return ucscNames ? G.name + "." + k.seq().name : k.seq().name; // Sequence::getFullName = genome.sequence
}

// halMafBlock.cpp:36-82
void MafExport::resetEntries() {
    reference = nullptr;
    refIndex = NULL_INDEX;
    for (auto i = entries.begin(); i != entries.end();) {
        auto next = i;
        ++next;
        MafBlockEntry *e = i->second;
        bool deleted = false;
        if (e->start == NULL_INDEX) {
            if (e->lastUsed > 10) {
                delete e;
                entries.erase(i);
                deleted = true;
            } else {
                ++e->lastUsed;
            }
        } else {
            e->lastUsed = 0;
        }
        if (!deleted) {
            e->start = NULL_INDEX;
            e->strand = '+';
            e->length = 0;
            e->sequence.clear();
        }
        i = next;
    }
}

// halMafBlock.cpp:84-112
void MafExport::initEntry(MafBlockEntry *entry, const SeqKey &k, const Dna *dna, bool clearSequence) {
    std::string sequenceName = getName(k);
    if (entry->name != sequenceName || k.g != entry->genome) {
        entry->name = sequenceName;
        entry->genome = k.g;
        entry->srcLength = k.seq().length;
    }
    if (dna) {
        entry->start = dna->pos - k.seq().start;
        entry->length = 0;
        entry->strand = dna->rev ? '-' : '+';
        if (dna->rev)
            entry->start = entry->srcLength - 1 - entry->start;
    } else {
        entry->start = NULL_INDEX;
        entry->length = 0;
        entry->strand = '+';
    }
    if (clearSequence)
        entry->sequence.clear();
    entry->tree = nullptr;
}

// halMafBlock.cpp:114-138
// ---- hal2maf --printTree: the tree of the column's bases (halMafBlock.cpp:121-292) ----
static void treeSetParent(MafTree *node, MafTree *parent) { // stTree_setParent: appended to the parent's children
    node->parent = parent;
    parent->children.push_back(node);
}
static bool treeEquals(const MafTree *a, const MafTree *b) { // stTree_equals
    if (a->label != b->label || a->children.size() != b->children.size())
        return false;
    for (size_t i = 0; i < a->children.size(); ++i)
        if (!treeEquals(a->children[i], b->children[i]))
            return false;
    return true;
}
static std::string treeNewick(const MafTree *t) { // stTree_getNewickTreeString without its final ';'
    std::string s;
    if (!t->children.empty()) {
        s += '(';
        for (size_t i = 0; i < t->children.size(); ++i) {
            if (i)
                s += ',';
            s += treeNewick(t->children[i]);
        }
        s += ')';
    }
    return s + t->label;
}
// :128-157: the node and its ancestors first among their siblings (first in a post-order walk)
static void prioritizeNodeInTree(MafTree *node) {
    MafTree *parent = node->parent;
    if (parent == nullptr)
        return;
    size_t nodeIndex = 0;
    while (parent->children[nodeIndex] != node)
        ++nodeIndex;
    std::swap(parent->children[0], parent->children[nodeIndex]);
    prioritizeNodeInTree(parent);
}
// :159-200
MafTree *MafExport::getTreeNode(const SegIt &segIt, bool modifyEntries) {
    MafTree *ret = new MafTree;
    const Genome &genome = alp->genomes[(size_t)segIt.g];
    const i64 position = segIt.getStartPosition();
    const SeqKey seq{alp, segIt.g, seqIndexBySite(genome, position)};
    Entries::const_iterator entryIt = entries.lower_bound(seq);
    if (entryIt != entries.end() && entryIt->first == seq) {
        MafBlockEntry *entry = nullptr;
        for (; entryIt != entries.end() && entryIt->first == seq; ++entryIt) {
            MafBlockEntry *curEntry = entryIt->second;
            i64 curEntryPos = curEntry->start + curEntry->length;
            if (curEntry->strand == '-')
                curEntryPos = curEntry->srcLength - 1 - curEntryPos;
            if (curEntryPos == position - seq.seq().start || curEntry->start == NULL_INDEX) {
                entry = curEntry;
                break;
            }
        }
        if (entry == nullptr) // (an assertion in the reference: the column was found appendable before its tree is built)
            throw std::runtime_error("printTree: no block entry continues at this base");
        ret->entry = entry;
        ret->label = entry->name;
        if (modifyEntries)
            entry->tree = ret;
    } else { // no entry for this sequence: an ancestor, ancestral sequence left out (--noAncestors)
        ret->label = genome.name;
    }
    return ret;
}
// :204-237: a node for every child segment (and its paralogs) of this bottom segment, and on down
void MafExport::buildTreeR(const SegIt &botIt, MafTree *node, bool modifyEntries) {
    const Genome &genome = botIt.G();
    for (size_t i = 0; i < genome.children.size(); ++i) {
        if (!botIt.hasChild((i64)i))
            continue;
        SegIt topIt;
        topIt.toChild(botIt, (i64)i);
        MafTree *canonicalParalog = getTreeNode(topIt, modifyEntries);
        treeSetParent(canonicalParalog, node);
        if (topIt.G().tBotParse[(size_t)topIt.idx] != NULL_INDEX) { // hasParseDown
            SegIt childBotIt;
            childBotIt.toParseDown(topIt);
            buildTreeR(childBotIt, canonicalParalog, modifyEntries);
        }
        if (topIt.hasNextParalogy()) { // the rest of the paralogy cycle hangs under the same parent node
            topIt.toNextParalogy();
            while (!topIt.isCanonicalParalog()) {
                MafTree *paralog = getTreeNode(topIt, modifyEntries);
                treeSetParent(paralog, node);
                if (topIt.G().tBotParse[(size_t)topIt.idx] != NULL_INDEX) {
                    SegIt childBotIt;
                    childBotIt.toParseDown(topIt);
                    buildTreeR(childBotIt, paralog, modifyEntries);
                }
                topIt.toNextParalogy();
            }
        }
    }
}
// :239-292: from any base of the column up to the segment that is the ancestor of all of them, then down
MafTree *MafExport::buildTree(ColumnIterator &col, bool modifyEntries) {
    const Dna *first = nullptr;
    for (auto c = col.colMap.begin(); c != col.colMap.end() && !first; ++c)
        if (!c->second.empty())
            first = &c->second[0];
    if (!first)
        throw std::runtime_error("printTree: empty column");
    const Genome &genome = alp->genomes[(size_t)first->g];
    SegIt topIt, botIt;
    bool haveBot = false;
    topIt.al = botIt.al = alp;
    if (genome.numTop == 0) { // the reference is the root genome
        botIt.g = first->g;
        botIt.top = false;
        botIt.toSite(first->pos);
        haveBot = true;
    } else {
        topIt.g = first->g;
        topIt.top = true;
        topIt.toSite(first->pos);
        while (topIt.hasParent()) {
            const int parent = topIt.G().parent;
            botIt.toParent(topIt);
            haveBot = true;
            if (alp->genomes[(size_t)parent].parent < 0 || botIt.G().bTopParse[(size_t)botIt.idx] == NULL_INDEX)
                break; // the root genome, or nothing above this segment
            SegIt up;
            up.toParseUp(botIt);
            topIt = up;
        }
    }
    MafTree *t;
    if (genome.numTop != 0 && !topIt.hasParent() && topIt.g == first->g && genome.numBot == 0) {
        t = getTreeNode(topIt, modifyEntries); // an insertion in a leaf: no bottom segment anywhere
    } else {
        // (the reference dereferences a null bottom iterator here when the column's first base is an insertion in a genome
        // that has bottom segments: undefined there, an error here)
        if (!haveBot)
            throw std::runtime_error("printTree: the column's first base has no parent in a genome with bottom segments");
        t = getTreeNode(botIt, modifyEntries);
        buildTreeR(botIt, t, modifyEntries);
    }
    return t;
}

void MafExport::updateEntry(MafBlockEntry *entry, const SeqKey *k, const Dna *dna) {
    if (dna != nullptr) {
        if (entry->start == NULL_INDEX)
            initEntry(entry, *k, dna, false);
        ++entry->length;
        entry->sequence.push_back(dnaBase(*alp, *dna));
    } else {
        entry->sequence.push_back('-');
    }
}

// halMafBlock.cpp:294-367
void MafExport::initBlock(ColumnIterator &col) {
    if (printTree && tree != nullptr) {
        delete tree;
        tree = nullptr;
    }
    resetEntries();
    Entries::iterator e = entries.begin();
    for (auto c = col.colMap.begin(); c != col.colMap.end(); ++c) {
        const SeqKey &sequence = c->first;
        if (c->second.empty()) {
            e = entries.lower_bound(sequence);
            if (e == entries.end() || e->first != sequence) {
                MafBlockEntry *entry = new MafBlockEntry;
                initEntry(entry, sequence, nullptr);
                e = entries.insert(Entries::value_type(sequence, entry));
            } else {
                initEntry(e->second, sequence, nullptr);
            }
        } else {
            for (auto d = c->second.begin(); d != c->second.end(); ++d) {
                if (e == entries.begin()) {
                    e = entries.lower_bound(sequence);
                    if (e == entries.end() || e->first != sequence)
                        e = entries.end();
                } else {
                    // the reference dereferences e before testing for end(); order swapped to stay defined
                    while (e != entries.end() && e->first != c->first)
                        ++e;
                }
                if (e == entries.end()) {
                    MafBlockEntry *entry = new MafBlockEntry;
                    initEntry(entry, sequence, &*d);
                    e = entries.insert(Entries::value_type(sequence, entry));
                } else {
                    initEntry(e->second, sequence, &*d);
                }
                ++e;
            }
        }
    }
    if (reference == nullptr) {
        const SeqKey referenceSequence = col.refSequenceKey();
        e = entries.lower_bound(referenceSequence);
        if (e == entries.end() || e->first != referenceSequence)
            e = entries.begin();
        reference = e->second;
        if (e->first == referenceSequence)
            refIndex = col.refSequencePosition();
    }
    if (printTree)
        tree = buildTree(col, true);
}

// halMafBlock.cpp:370-395
void MafExport::appendColumn(ColumnIterator &col) {
    Entries::iterator e = entries.begin();
    for (auto c = col.colMap.begin(); c != col.colMap.end(); ++c) {
        const SeqKey &sequence = c->first;
        for (auto d = c->second.begin(); d != c->second.end(); ++d) {
            while (e != entries.end() && e->first != sequence) {
                updateEntry(e->second, nullptr, nullptr);
                ++e;
            }
            updateEntry(e->second, &sequence, &*d);
            ++e;
        }
    }
    for (; e != entries.end(); ++e)
        updateEntry(e->second, nullptr, nullptr);
}

// halMafBlock.cpp:401-452
bool MafExport::canAppendColumn(ColumnIterator &col) {
    Entries::iterator e = entries.begin();
    for (auto c = col.colMap.begin(); c != col.colMap.end(); ++c) {
        const SeqKey &sequence = c->first;
        i64 sequenceStart = sequence.seq().start;
        for (auto d = c->second.begin(); d != c->second.end(); ++d) {
            while (e != entries.end() && e->first != sequence)
                ++e;
            if (e == entries.end())
                return false;
            MafBlockEntry *entry = e->second;
            if (entry->start != NULL_INDEX) {
                if (entry->length >= maxBlockLength || (entry->length > 0 && (entry->strand == '-') != d->rev))
                    return false;
                i64 pos = d->pos - sequenceStart;
                if (d->rev)
                    pos = entry->srcLength - 1 - pos;
                if (pos - entry->start != entry->length)
                    return false;
            }
            ++e;
        }
    }
    if (printTree) { // :443-448: the column's tree must be the block's
        MafTree *t = buildTree(col, false);
        const bool ret = treeEquals(t, tree);
        delete t;
        return ret;
    }
    return true;
}

bool MafExport::referenceIsAllGaps() const { // halMafBlock.h:95-102,136-138
    if (reference == nullptr)
        return false;
    for (char c : reference->sequence)
        if (c != '-')
            return false;
    return true;
}

static void printEntry(std::ostream &os, const MafBlockEntry &e) { // halMafBlock.cpp:454-458
    os << "s\t" << e.name << '\t' << e.start << '\t' << e.length << '\t' << e.strand << '\t' << e.srcLength << '\t' << e.sequence << '\n';
}

// halMafBlock.cpp:499-519
static void printTreeEntries(const MafTree *t, std::ostream &os) { // :472-483: post order
    for (const MafTree *c : t->children)
        printTreeEntries(c, os);
    if (t->entry != nullptr) // (null for an ancestor left out by --noAncestors)
        printEntry(os, *t->entry);
}

void MafExport::printBlock(std::ostream &os) const {
    if (printTree) { // printBlockWithTree, :485-497
        if (reference->tree != nullptr)
            prioritizeNodeInTree(reference->tree); // the reference first
        os << "a tree=\"" << treeNewick(tree) << ";\"\n";
        printTreeEntries(tree, os);
        return;
    }
    os << "a\n";
    if (reference->start == NULL_INDEX) {
        if (refIndex != NULL_INDEX) {
            reference->start = refIndex;
            printEntry(os, *reference);
            reference->start = NULL_INDEX;
        }
    } else {
        printEntry(os, *reference);
    }
    for (auto e = entries.begin(); e != entries.end(); ++e)
        if (e->second->start != NULL_INDEX && e->second != reference)
            printEntry(os, *e->second);
}

// halMafExport.cpp:15-88
void MafExport::convertSequence(std::ostream &os, const Alignment &al, int genome, int seqIdx, i64 startPosition, i64 length,
                                const std::set<int> &targets) {
    alp = &al;
    const Sequence &seq = al.genomes[(size_t)genome].seqs[(size_t)seqIdx];
    if (startPosition >= seq.length || startPosition + length > seq.length)
        throw std::runtime_error("Invalid range specified for convertGenome");
    if (length == 0)
        length = seq.length - startPosition;
    if (length == 0)
        throw std::runtime_error("Cannot convert zero length sequence");
    i64 lastPosition = startPosition + (length - 1);
    if (!append && !headerWritten) { // writeHeader: only when nothing has been written to the stream yet (:15-23)
        os << "##maf version=1 scoring=N/A\n"
           << "# hal " << al.newick << std::endl
           << std::endl;
        headerWritten = true;
    }
    auto t0 = std::chrono::steady_clock::now();
    // Sequence::getColumnIterator (api/mmap_impl/mmapSequence.cpp:41-52): sequence-relative -> genome coordinates
    ColumnIterator colIt(&al, genome, &targets, startPosition + seq.start, lastPosition + seq.start, noDupes, noAncestors, onlyOrthologs,
                         unique, maxRefGap);
    size_t appendCount = 0;
    if (!unique || colIt.isCanonicalOnRef()) {
        initBlock(colIt);
        appendColumn(colIt);
        ++appendCount;
    }
    ++numColumns;
    size_t numBlocks = 0;
    while (!colIt.lastColumn()) {
        colIt.toRight();
        ++numColumns;
        if (!unique || colIt.isCanonicalOnRef()) {
            if (appendCount == 0)
                initBlock(colIt);
            if (!canAppendColumn(colIt)) {
                if (numBlocks++ % 1000 == 0)
                    colIt.defragment();
                if (appendCount > 0 && (keepEmptyRefBlocks || !referenceIsAllGaps())) {
                    printBlock(os);
                    os << '\n';
                }
                initBlock(colIt);
            }
            appendColumn(colIt);
            ++appendCount;
        }
    }
    if (appendCount > 0 && (keepEmptyRefBlocks || !referenceIsAllGaps())) {
        printBlock(os);
        os << std::endl;
    }
    seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// halMafExport.cpp:90-153
void MafExport::convertEntireAlignment(std::ostream &os, const Alignment &al) {
    alp = &al;
    size_t appendCount = 0, numBlocks = 0;
    if (!headerWritten) {
        os << "##maf version=1 scoring=N/A\n"
           << "# hal " << al.newick << std::endl
           << std::endl;
        headerWritten = true;
    }
    // getLeafGenomes (api/impl/halCommon.cpp:197-207) over Alignment::getLeafNamesBelow (api/mmap_impl/mmapAlignment.h:155-171)
    std::vector<int> leafGenomes;
    {
        std::deque<int> bfQueue;
        bfQueue.push_front(al.root());
        while (!bfQueue.empty()) {
            const int current = bfQueue.back();
            const std::vector<int> &children = al.genomes[(size_t)current].children;
            if (children.empty() && current != al.root())
                leafGenomes.push_back(current);
            for (size_t i = 0; i < children.size(); ++i)
                bfQueue.push_front(children[i]);
            bfQueue.pop_back();
        }
    }
    std::map<int, PositionCache> visitCache;
    for (size_t i = 0; i < leafGenomes.size(); i++) {
        const int genome = leafGenomes[i];
        const Genome &G = al.genomes[(size_t)genome];
        if (G.totalLength == 0)
            continue;
        ColumnIterator colIt(&al, genome, nullptr, 0, G.totalLength - 1, noDupes, noAncestors, onlyOrthologs, true, 0);
        colIt.visitCache = visitCache; // setVisitCache: the iterator's own cache (its constructor's first column) is dropped
        // toSite(0, length - 1) (halColumnIterator.cpp:146-165)
        colIt.seqIdx = 0;
        while (colIt.seqIdx + 1 < (int)G.seqs.size() && G.seqs[(size_t)colIt.seqIdx].length == 0)
            ++colIt.seqIdx;
        colIt.refSeqIdx = colIt.seqIdx;
        colIt.upper.clear();
        colIt.defragment();
        colIt.firstIndex = colIt.index = 0;
        colIt.lastIndex = G.totalLength - 1;
        colIt.toRight();
        for (;;) {
            if (appendCount == 0)
                initBlock(colIt);
            if (!canAppendColumn(colIt)) {
                if (numBlocks++ % 1000 == 0)
                    colIt.defragment();
                if (appendCount > 0) {
                    printBlock(os);
                    os << '\n';
                }
                initBlock(colIt);
            }
            appendColumn(colIt);
            appendCount++;
            ++numColumns;
            if (colIt.lastColumn())
                break;
            colIt.toRight();
        }
        visitCache = colIt.visitCache;
    }
    if (appendCount > 0) {
        printBlock(os);
        os << std::endl;
    }
}

} // namespace orc
