// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).
// Restatement of the column engine: api/impl/halColumnIterator.cpp — the configuration of hal2maf's and halAlignmentDepth's
// defaults (maxInsertLength == 0, unique == false: columns are independent, halColumnIterator.cpp:785-787), --unique (the visit
// cache) and maxInsertLength > 0 (hal2maf --maxRefGap: the stack of inserted / deleted ranges the iterator walks between two
// reference columns, :65-144, :357-401, api/inc/halColumnIteratorStack.h, with Rearrangement's deletion and insertion cycles in
// the atomic, gap-threshold-0 form the iterator uses them in, api/impl/halRearrangement.cpp:133-160, 386-545) —,
// maf/impl/halMafBlock.cpp, maf/impl/halMafExport.cpp and alignmentDepth/halAlignmentDepth.cpp:215-347.
#pragma once
#include "oracle_mapper.hpp"
#include <map>
#include <ostream>

namespace orc {

// DnaIterator state (api/inc/halDnaIterator.h): genome position + strand
struct Dna {
    int g;
    i64 pos;
    bool rev;
};

// api/inc/halColumnIterator.h:45-50
struct SeqKey {
    const Alignment *al;
    int g, s;
    bool operator<(const SeqKey &o) const {
        int diff = al->genomes[(size_t)g].name.compare(al->genomes[(size_t)o.g].name);
        return diff < 0 || (diff == 0 && s < o.s);
    }
    bool operator==(const SeqKey &o) const {
        return g == o.g && s == o.s;
    }
    bool operator!=(const SeqKey &o) const {
        return !(*this == o);
    }
    const Sequence &seq() const {
        return al->genomes[(size_t)g].seqs[(size_t)s];
    }
};

// api/impl/halPositionCache.cpp:12-78: set of positions kept as merged intervals (map: last -> first)
struct PositionCache {
    std::map<i64, i64> set;
    bool find(i64 pos) const {
        auto i = set.lower_bound(pos);
        return i != set.end() && i->second <= pos;
    }
    bool insert(i64 pos) { // false if already present
        if (find(pos))
            return false;
        i64 first = pos, last = pos;
        auto right = set.lower_bound(pos); // interval starting after pos (its first == pos + 1 merges)
        if (right != set.end() && right->second == pos + 1) {
            last = right->first;
            set.erase(right);
        }
        auto left = set.find(pos - 1); // interval ending at pos - 1 merges
        if (left != set.end()) {
            first = left->second;
            set.erase(left);
        }
        set[last] = first;
        return true;
    }
};

struct ColumnIterator {
    typedef std::vector<Dna> DNASet;
    typedef std::map<SeqKey, DNASet> ColumnMap;

    // ColumnIteratorStack::Entry (halColumnIteratorStack.h:47-107) without its linked iterators (they only save the search for
    // the site: every column is located afresh here)
    struct Entry {
        int g = -1, seqIdx = -1;
        i64 firstIndex = 0, index = 0, lastIndex = 0, cumulativeSize = 0;
        bool reversed = false;
        bool pastEnd() const {
            return reversed ? index < firstIndex : index > lastIndex;
        }
    };
    const Alignment *al;
    int refGenome;
    bool noDupes, noAncestors, onlyOrthologs, unique = false;
    i64 maxInsertLength = 0;
    std::map<int, PositionCache> visitCache; // halColumnIterator.h: VisitCache (per genome)
    bool brk = false;                        // _break
    i64 leftmostRefPos = 0;
    std::set<int> targets, scope;
    // _stack: entry 0 (the reference range; its fields keep their names from the one-entry case) and what lies on top of it;
    // _insertionStack, _deletionStack: filled while a column is walked, moved onto _stack behind it (halColumnIterator.cpp:121-123)
    Entry base;
    int &seqIdx = base.seqIdx;
    i64 &firstIndex = base.firstIndex, &index = base.index, &lastIndex = base.lastIndex;
    std::vector<Entry> upper, insertionStack, deletionStack;
    ColumnMap colMap;
    int prevRefSeq;
    i64 prevRefIndex;
    int refSeqIdx; // _ref: the reference sequence (follows entry 0 across sequence boundaries)

    ColumnIterator(const Alignment *a, int reference, const std::set<int> *tgts, i64 columnIndex, i64 lastColumnIndex, bool noDupes_,
                   bool noAncestors_, bool onlyOrthologs_, bool unique_ = false, i64 maxInsertLength_ = 0);
    ColumnIterator(const ColumnIterator &) = delete;
    bool isCanonicalOnRef() const { // halColumnIterator.cpp:210-214
        return leftmostRefPos >= firstIndex && leftmostRefPos <= lastIndex;
    }
    void toRight();
    bool lastColumn() const { // halColumnIterator.cpp:167-169
        return upper.empty() && base.pastEnd();
    }
    Entry &top() {
        return upper.empty() ? base : upper.back();
    }
    bool topInBounds() {
        const Entry &e = top();
        return e.index >= e.firstIndex && e.index <= e.lastIndex;
    }
    void defragment();
    SeqKey refSequenceKey() const {
        return SeqKey{al, refGenome, prevRefSeq};
    }
    i64 refSequencePosition() const {
        return prevRefIndex;
    }

  private:
    bool parentInScope(int g) const {
        return scope.empty() || scope.count(al->genomes[(size_t)g].parent);
    }
    bool childInScope(int g, i64 slot) const {
        return scope.empty() || scope.count(al->genomes[(size_t)g].children[(size_t)slot]);
    }
    void recursiveUpdate();
    bool colMapInsert(const SegIt &it);
    void nextFreeIndex();
    void updateParent(const SegIt &top);
    void updateChild(const SegIt &bot, i64 slot);
    void updateNextTopDup(const SegIt &top);
    void updateParseUp(const SegIt &bot);
    void updateParseDown(const SegIt &top);
    bool handleDeletion(const SegIt &inputTop);
    bool handleInsertion(const SegIt &inputTop);
    static void pushEntry(std::vector<Entry> &st, int g, int seqIdx, i64 index, i64 lastIndex, bool reversed);
};

char dnaBase(const Alignment &al, const Dna &d); // DnaIterator::getBase, halDnaIterator.h:131-138

// alignmentDepth/halAlignmentDepth.cpp:318-347 printGenome / :215-308 printSequence
void printDepthGenome(std::ostream &os, const Alignment &al, int genome, int sequence /* -1 = all */, const std::set<int> &targetSet,
                      i64 start, i64 length, i64 step, bool countDupes, bool noAncestors);

// maf/impl/halMafBlock.cpp + halMafExport.cpp
// the part of sonLib's stTree the MAF block uses (hal2maf --printTree; sonLib is not in the reference tree: restated from its
// use in maf/impl/halMafBlock.cpp:121-292, 472-496 and its published behaviour — children in the order they were given a parent,
// Newick text "(child,child)label;" without lengths when none was set, equality = same label and equal children in order)
struct MafBlockEntry;
struct MafTree {
    MafTree *parent = nullptr;
    std::vector<MafTree *> children;
    std::string label;
    MafBlockEntry *entry = nullptr; // client data
    ~MafTree() {
        for (MafTree *c : children)
            delete c;
    }
};
struct MafBlockEntry {
    int genome = -1;
    std::string name;
    i64 start = NULL_INDEX, length = 0, srcLength = 0;
    char strand = '+';
    std::string sequence;
    short lastUsed = 0;
    MafTree *tree = nullptr;
};

struct MafExport {
    bool noDupes = false, noAncestors = false, ucscNames = true /* Genome.Sequence */, onlyOrthologs = false, keepEmptyRefBlocks = false,
         append = false, unique = false;
    i64 maxBlockLength = 1000; // MafBlock::defaultMaxLength, halMafBlock.cpp:16
    i64 maxRefGap = 0;         // MafExport::_maxRefGap = the column iterator's maxInsertLength (halMafExport.cpp:47)
    bool printTree = false;    // hal2maf --printTree
    void convertSequence(std::ostream &os, const Alignment &al, int genome, int seq, i64 startPosition, i64 length,
                         const std::set<int> &targets);
    // maf/impl/halMafExport.cpp:90-153 (hal2maf --global)
    void convertEntireAlignment(std::ostream &os, const Alignment &al);
    size_t numColumns = 0;
    double seconds = 0;

  private:
    typedef std::multimap<SeqKey, MafBlockEntry *> Entries;
    Entries entries;
    MafBlockEntry *reference = nullptr;
    i64 refIndex = NULL_INDEX;
    bool headerWritten = false;
    const Alignment *alp = nullptr;
    std::string getName(const SeqKey &k) const;
    void resetEntries();
    void initEntry(MafBlockEntry *e, const SeqKey &k, const Dna *dna, bool clearSequence = true);
    void updateEntry(MafBlockEntry *e, const SeqKey *k, const Dna *dna);
    void initBlock(ColumnIterator &col);
    void appendColumn(ColumnIterator &col);
    bool canAppendColumn(ColumnIterator &col);
    void printBlock(std::ostream &os) const;
    bool referenceIsAllGaps() const;
    MafTree *tree = nullptr;
    MafTree *getTreeNode(const SegIt &segIt, bool modifyEntries);
    void buildTreeR(const SegIt &botIt, MafTree *node, bool modifyEntries);
    MafTree *buildTree(ColumnIterator &col, bool modifyEntries);

  public:
    ~MafExport();
};

} // namespace orc
