// Host adapters with the reference's interfaces for the column engine: MafExport (maf/inc/halMafExport.h:30-68)
// and halAlignmentDepth's printGenome (alignmentDepth/halAlignmentDepth.cpp:318-347).  Column contents come from
// the GPU (hgx_columns.hip); block assembly and text formatting are sequential by definition in the reference
// (the block state depends on every earlier column) and stay on the host.
#pragma once
#include "hgx_columns_engine.hpp"
#include <deque>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <future>
#include <map>
#include <ostream>
#include <set>

namespace hgx {

// A stream buffer that can give room for a stretch of text at once (the C ABI's output buffer, hgx_capi.cpp): hal2maf's rendering
// threads copy their parts of a batch into it side by side instead of one after the other through ostream::write.
struct BulkSink {
    virtual char *room(size_t n) = 0; // n bytes at the end of the text, counted as written; null: no room (write them the usual way)
    virtual ~BulkSink() {}
};


// halAlignmentDepth: writes the wig text of printGenome.  sequence = -1: all sequences of the genome.
void alignmentDepth(std::ostream &os, hgx_alignment *h, int genome, int sequence, const std::set<int> &targetSet, int64_t start,
                    int64_t length, int64_t step, bool countDupes, bool noAncestors, ColumnStats *stats = nullptr,
                    const std::vector<hgx_alignment *> *moreDevices = nullptr);
// hal2mafMP.py's slicing (maf/hal2mafMP.py:63-79, 176-190): the range of every reference sequence cut into slices of sliceSize
// columns (0: the range divided evenly over the handles), every slice an export of its own, the texts put together in order
// with the header of the first only.  handles: device clones of one alignment (hgx_clone_to_device); the slices are dealt to
// them as they become free, one host thread per handle.
struct MafExportSettings {
    bool noDupes = false, noAncestors = false, ucscNames = true, onlyOrthologs = false, keepEmptyRefBlocks = false, unique = false,
         printTree = false;
    int64_t maxBlockLength = 1000, maxRefGap = 0;
};
void mafExportSliced(std::ostream &os, const std::vector<hgx_alignment *> &handles, int genome, int sequence, int64_t start, int64_t length,
                     int64_t sliceSize, const MafExportSettings &cfg, const std::set<int> &targets);

class MafExport {
  public:
    // maf/inc/halMafExport.h:36-58
    void setNoDupes(bool v) { _noDupes = v; }
    // the columns the handle serves in all (hgx_maf_export_multi's slices are exports of their own: the per-base tracks of
    // hgx_maf_kernels.hpp are kept with the handle and pay over all of them)
    void setExportHint(int64_t columns) { _exportHint = columns; }
    void setNoAncestors(bool v) { _noAncestors = v; }
    void setUcscNames(bool v) { _ucscNames = v; }
    void setAppend(bool v) { _append = v; }
    // stored as it is, like MafBlock::setMaxLength (maf/inc/halMafBlock.h:133): with 0 (or less) every column starts a block
    void setMaxBlockLength(int64_t v) { _maxBlockLength = v < 0 ? 0 : v; }
    void setOnlyOrthologs(bool v) { _onlyOrthologs = v; }
    void setKeepEmptyRefBlocks(bool v) { _keepEmptyRefBlocks = v; }
    void setMaxRefGap(int64_t v) { _maxRefGap = v; }
    void setUnique(bool v) { _unique = v; }
    // hal2maf --printTree: every block with the tree of its rows (halMafBlock.cpp:121-292, 485-497); blocks also end where the
    // column's tree changes (:443-448)
    void setPrintTree(bool v) { _printTree = v; }
    // maf/impl/halMafExport.cpp:25-88; positions are sequence-relative, length 0 = to the end
    void convertSequence(std::ostream &mafStream, hgx_alignment *alignment, int genome, int seq, int64_t startPosition, int64_t length,
                         const std::set<int> &targets);
    // hal2maf --global: every column of the alignment once (maf/impl/halMafExport.cpp:90-153)
    void convertEntireAlignment(std::ostream &mafStream, hgx_alignment *alignment);
    // hal2maf --refTargets: MafBed::visitLine (maf/impl/halMafBed.cpp:24-52) over a BED stream of reference intervals
    void convertBed(std::ostream &mafStream, hgx_alignment *alignment, int genome, std::istream &bedStream, const std::set<int> &targets);
    ~MafExport();
    ColumnStats stats;
    size_t chunkColumns = 1u << 21;

  private:
    // the part of sonLib's stTree the block uses (sonLib is not in the reference tree; restated from its use in halMafBlock.cpp and
    // its published behaviour: children in the order they were given their parent, Newick text "(child,child)label;" without
    // lengths when none was set, two trees equal when labels and children, in order, are)
    struct Entry;
    struct Tree {
        Tree *parent = nullptr;
        std::vector<Tree *> children;
        std::string label;
        Entry *entry = nullptr;
        ~Tree() {
            for (Tree *c : children)
                delete c;
        }
    };
    struct Entry { // MafBlockEntry, maf/inc/halMafBlock.h:60-118
        int genome = -1;
        std::string name;
        int64_t start = NULL_INDEX, length = 0, srcLength = 0;
        char strand = '+';
        std::string sequence;
        short lastUsed = 0;
        // run mode (the default export path): the row's text is not built column by column; the entry keeps the runs it
        // consists of and the text is rendered from the packed DNA when the block is printed
        struct Seg {
            int64_t pos; // genome coordinate of the run's first base (kind 2: the run walks left from there)
            int32_t n;
            uint8_t kind; // 0 gap, 1 forward bases, 2 reverse-strand bases (complemented)
        };
        std::vector<Seg> segs;
        uint32_t nameId = 0; // index into _names (run mode snapshots refer to names by id)
        Tree *tree = nullptr;
    };
    struct Key {
        int rank, genome, seq;
        bool operator<(const Key &o) const { return rank < o.rank; }
    };
    typedef std::multimap<Key, Entry *> Entries;
    typedef std::map<Key, std::vector<const ColumnRowHost *>> ColumnMap;
    bool _noDupes = false, _noAncestors = false, _ucscNames = true, _append = false, _onlyOrthologs = false, _keepEmptyRefBlocks = false,
         _unique = false, _printTree = false;
    int64_t _exportHint = 0;
    Tree *_tree = nullptr;
    Tree *getTreeNode(int genome, int64_t pos, bool modifyEntries);
    void buildTreeR(int genome, int64_t bottomSegment, int64_t pos, Tree *node, bool modifyEntries);
    Tree *buildTree(const ColumnMap &col, bool modifyEntries);
    int64_t _maxBlockLength = 1000, _maxRefGap = 0;
    Entries _entries;
    Entry *_reference = nullptr;
    int64_t _refIndex = NULL_INDEX;
    hgx_alignment *_al = nullptr;
    std::vector<std::vector<int>> _rank; // [genome][sequence] -> order of ColumnIterator::SequenceLess
    std::vector<int> _rankGenome, _rankSeq; // and back
    void buildRanks();
    Key keyOf(int genome, int64_t pos) const;
    void resetEntries();
    void initEntry(Entry *e, const Key &k, const ColumnRowHost *row, bool clearSequence = true);
    void updateEntry(Entry *e, const Key *k, const ColumnRowHost *row);
    void initBlock(const ColumnMap &col, const Key &refKey, int64_t refPos);
    void appendColumn(const ColumnMap &col);
    bool canAppendColumn(const ColumnMap &col);
    bool referenceIsAllGaps() const;
    void printBlock(std::ostream &os) const;
    // run mode: blocks are snapshotted (rows as runs) and rendered to text in batches by several threads
    bool _segMode = false;
    struct RowSnap {
        uint32_t nameId, firstSeg, numSegs;
        int64_t start, length, srcLength;
        int32_t genome;
        char strand;
    };
    struct BlockSnap {
        uint32_t firstRow, numRows;
    };
    std::vector<BlockSnap> _snapBlocks;
    std::vector<RowSnap> _snapRows;
    std::vector<Entry::Seg> _snapSegs;
    std::deque<std::string> _names;               // every row name handed out so far (a deque: rendering threads keep references)
    std::map<std::string, uint32_t> _nameIds;
    void snapshotBlock();
    void flushSnapshots(std::ostream &os);
    std::future<void> _pendingWrite;              // rendering + writing of the previous batch, running beside the state machine
    // the run-compressed export renders several batches at a time (RunMachine::flush): every batch's threads make its text apart,
    // then take their turn — by ticket, in the order of the batches — to be given room in the output and copy it there
    struct WriteOrder {
        std::mutex mu;
        std::condition_variable cv;
        size_t next = 0; // the ticket whose batch may write
        void wait(size_t ticket) {
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&]() { return next == ticket; });
        }
        std::set<size_t> gone; // (batches that ended before their turn came: an exception)
        void done(size_t ticket) {
            std::lock_guard<std::mutex> lock(mu);
            if (next == ticket) {
                ++next;
                while (gone.erase(next))
                    ++next;
            } else if (ticket > next) {
                gone.insert(ticket);
            }
            cv.notify_all();
        }
    };
    std::shared_ptr<WriteOrder> _writeOrder = std::make_shared<WriteOrder>();
    size_t _ticketsIssued = 0;
    std::deque<std::future<void>> _pendingWrites;
    void waitPendingWrite() {
        if (_pendingWrite.valid())
            _pendingWrite.get();
        std::exception_ptr first;
        while (!_pendingWrites.empty()) { // (every one is waited for, whatever the ones before threw)
            try {
                _pendingWrites.front().get();
            } catch (...) {
                if (!first)
                    first = std::current_exception();
            }
            _pendingWrites.pop_front();
        }
        if (first)
            std::rethrow_exception(first);
    }
    void appendRun(Entry *e, const ColumnRowHost *row, int64_t pos, int64_t n);
    // ---- the run-compressed export (hgx_columns_host.cpp: RunMachine): MafBlock's state on flat arrays ----
    struct RunMachine;
    friend struct RunMachine;
    void convertSequenceRuns(std::ostream &mafStream, hgx_alignment *alignment, int genome, int seq, int64_t startPosition, int64_t length,
                             const ColumnOptions &opt);
    // the same walk over slices of the export side by side (hgx_columns_host.cpp); batches: the export's batches as they arrive (the walk
    // begins when the first two are there); false: fewer than two batches — nothing was written, the caller walks with one thread
    struct Arrivals;
    bool walkSliced(std::ostream &mafStream, Arrivals &batches, int refRank, int64_t startPosition, size_t &numBlocks, int *rounds = nullptr,
                    unsigned *threads = nullptr, double *seconds = nullptr);
    // --maxRefGap > 0: the column iterator with its stack of inserted / deleted ranges (halColumnIterator.cpp:65-144, 357-405),
    // replayed over the columns and indel events the device returns (hgx_gap_kernels.hpp)
    void convertSequenceGapped(std::ostream &mafStream, hgx_alignment *alignment, int genome, int seq, int64_t first, int64_t last,
                               const ColumnOptions &opt);
};

// the last run-compressed hal2maf export of this process, as a JSON object ("null" before the first): who walked it, how long the parts took
std::string mafLastExportInfo();

} // namespace hgx
