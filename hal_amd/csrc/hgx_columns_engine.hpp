// Internal C++ interface of the column engine (implemented in hgx_columns.hip).
#pragma once
#include "../../include/hgx.h"
#include "hgx_device.hpp"
#include <functional>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace hgx {

struct ColumnRowHost { // mirrors ColumnRow of hgx_column_kernels.hpp
    int64_t pos;
    int32_t genome;
    uint8_t rev;
    char base;
    uint8_t _pad[2];
};

// Host memory the device's large copies land in: page-locked blocks (hipHostMalloc) kept in a small pool.  A copy into pageable
// memory goes through the runtime's staging buffers at a fraction of the link's rate, and a fresh vector of a batch's rows is
// cleared and page-faulted before it is overwritten: a config-3 export moves half a gigabyte of rows this way.  Requests below a
// megabyte, and every request where there is no device (the profiling build's replay), are ordinary memory.
void *hostBlockTake(size_t bytes); // never null (std::bad_alloc)
void hostBlockGive(void *p) noexcept;
template <class T> struct HostBlockAllocator {
    typedef T value_type;
    HostBlockAllocator() = default;
    template <class U> HostBlockAllocator(const HostBlockAllocator<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(hostBlockTake(n * sizeof(T))); }
    void deallocate(T *p, size_t) noexcept { hostBlockGive(p); }
    template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; } // (not cleared: a copy fills it)
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) {
        ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...);
    }
    template <class U> bool operator==(const HostBlockAllocator<U> &) const { return true; }
    template <class U> bool operator!=(const HostBlockAllocator<U> &) const { return false; }
};

struct ColumnOptions {
    bool noDupes = false, noAncestors = false, onlyOrthologs = false;
    std::vector<int> targets; // genome ids; empty = everything (halColumnIterator.cpp:45-51)
};

typedef std::vector<ColumnRowHost, HostBlockAllocator<ColumnRowHost>> HeadRows; // the heads' rows of a batch of columns

// thrown by columnsHeadRowsHost when a chunk of columns holds 2^32 rows or more (its offsets are 32-bit): the caller halves
// the chunk and asks again
struct ColumnChunkTooLarge : std::runtime_error {
    ColumnChunkTooLarge() : std::runtime_error("a chunk of columns holds 2^32 rows or more") {}
};

struct ColumnStats {
    double depth_ms = 0, rows_ms = 0;
    uint64_t columns = 0, rows = 0;
    uint64_t top_derefs = 0, bottom_derefs = 0; // segment records the walks logically dereferenced (depth kernel only)
    uint64_t sweep_bytes = 0;                   // depth by tree sweeps: bytes of the per-base tracks
};

// per-column values of halAlignmentDepth: mode 0 = distinct genomes - 1, 1 = bases - 1 (--countDupes), 2 = bases
// first: genome coordinate; results for columns first, first+step, ... (count of them)
void columnsDepthHost(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                      int32_t *out, ColumnStats *stats, bool countDerefs = false);
// same, results left on the device (d_out: device int32[count]); stream = hipStream_t
void columnsDepthDevice(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                        int32_t *d_out, void *stream, ColumnStats *stats, bool countDerefs = false, bool perBase = false);
// the same values handed to `sink(values, index of the first, how many)` chunk by chunk, in order, from page-locked blocks of `chunk`
// values: the copy of a chunk goes on while the sink has the chunk before (halAlignmentDepth's text: no array of a genome's length
// on the host)
void columnsDepthChunksHost(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                            ColumnStats *stats, int64_t chunk, const std::function<void(const int32_t *, int64_t, int64_t)> &sink);
// the same values as halAlignmentDepth's lines ("%d\n" each), made on the device (k_wig_text) and handed to `sink(text, bytes)` chunk by
// chunk, in order
void columnsDepthTextChunksHost(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                                ColumnStats *stats, int64_t chunk, const std::function<void(const char *, size_t)> &sink);
// every reported base of columns [first, first+count), in the reference's ColumnMap insertion order;
// rowOffset gets count+1 entries
void columnsRowsHost(hgx_alignment *h, int ref, int64_t first, int64_t count, const ColumnOptions &opt, bool withDna,
                     std::vector<uint64_t> &rowOffset, std::vector<ColumnRowHost> &rows, ColumnStats *stats);

// The columns of a ColumnIterator with maxInsertLength > 0 (hgx_gap_kernels.hpp): column i starts from base asks[i] — any genome,
// either orientation — and comes back as EVERY base its walk visits, in the reference's order, with what the host's replay of the
// iterator's sequential state needs: _pad[0] & 7 = 0 a reported base, 1 a base colMapInsert's filters keep out of the column, 2 / 3
// a deleted / inserted range the walk met behind the previous base (pos = its first base, genome, rev = the orientation it is
// to be walked in; the next row, kind 4, carries its last base); _pad[0] & 8: the base was reached by updateParent; _pad[1]: level
// in the walk's upward chain.  ref: the iterator's reference genome (scope and targets are relative to it).
struct GapAskHost {
    int64_t pos;
    int32_t genome;
    int32_t reversed;
};
// _pad[0] & 16: a paralog inserted by updateNextTopDup's loop; _pad[1] of a base row: its depth in the recursion (the level, on
// the upward chain).  events = false: no deleted / inserted ranges are looked for (maxInsertLength == 0).
void columnsGapRowsHost(hgx_alignment *h, int ref, const std::vector<GapAskHost> &asks, const ColumnOptions &opt, bool withDna,
                        std::vector<uint64_t> &rowOffset, std::vector<ColumnRowHost> &rows, ColumnStats *stats, bool events = true);

// Run-compressed form for the MAF writer: head[c] == 1 when column c does not simply continue column c-1 (same rows
// advanced by one base); only heads have their rows returned (headOffset: one entry per head, + 1).
// uniqueFirst >= 0: hal2maf --unique over a range that begins at genome coordinate uniqueFirst (hgx_column_kernels.hpp:
// k_column_unique_count) — head[c] == 2: column c is not walked by the reference's iterator (no rows, no entry in headOffset),
// head[c] == 3: walked but not written (its rows are returned: their sequences become keys of the column map); a written
// column is a head or a continuation of the written column before it, as above.
void columnsHeadRowsHost(hgx_alignment *h, int ref, int64_t first, int64_t count, const ColumnOptions &opt, bool withDna,
                         std::vector<uint8_t> &head, std::vector<uint32_t> &headOffset, HeadRows &headRows, ColumnStats *stats,
                         int64_t uniqueFirst = -1, int64_t exportColumns = 0);
// (exportColumns: the columns of the whole export this chunk belongs to — the plain export takes its heads from per-base tracks
// made by sweeps over whole genomes, hgx_maf_kernels.hpp, when the export is long enough for them to pay; HGX_MAF_SWEEP=1 / 0
// forces / forbids them)

// The batches of a plain hal2maf export as a stream (hgx_columns.hip: MafChunkStream): chunks are submitted ahead and collected in
// order; a chunk comes back as the walk wants it — the head marks, the heads' columns and row offsets, their rows described (key,
// rank of the sequence) and sorted the way the column map holds them.  rankBase[g]: the rank of genome g's first sequence.  Open:
// null when the export is not one for the per-base tracks (then columnsHeadRowsHost serves it).  Collect: false when the chunk has to
// be made by columnsHeadRowsHost after all (no room, 2^32 rows, tracks that differ from the walk) — it and everything behind it.
struct MafChunkRow { // = RunMachine::PRow of hgx_columns_host.cpp
    int64_t key;
    int32_t rank;
    uint32_t ord;
};
struct MafChunkOut {
    int64_t n = 0;
    std::vector<uint8_t> head;
    std::vector<uint32_t> headOff, headCol;
    MafChunkRow *rows = nullptr; // a page-locked block (hostBlockGive it)
    size_t numRows = 0;
};
struct MafChunkStream;
MafChunkStream *mafChunkStreamOpen(hgx_alignment *h, int ref, const ColumnOptions &opt, const std::vector<int32_t> &rankBase, int64_t maxChunk,
                                   int64_t exportColumns, ColumnStats *stats, int64_t uniqueFirst = -1);
// (uniqueFirst >= 0: hal2maf --unique over a range that begins at that genome coordinate — marks 2 and 3 as columnsHeadRowsHost's)
void mafChunkStreamSubmit(MafChunkStream *M, int64_t first, int64_t count);
bool mafChunkStreamCollect(MafChunkStream *M, MafChunkOut &out);
size_t mafChunkStreamInFlight(const MafChunkStream *M);
void mafChunkStreamClose(MafChunkStream *M);

// hal2maf's text on the device (hgx_maf_render_kernels.hpp): the walk's log of a batch of blocks — every array in host memory, in the
// layouts of that header — becomes the batch's MAF text, which comes back in a page-locked block (hostBlockGive it).  false: not on
// this handle or at this size (no device, the bases not on it, a batch of four gigabytes of text): the caller renders it itself.
struct MafRenderBlock { // RunMachine::BlockLog + where the block's entries' slots begin
    uint32_t firstEnt, numEnts, firstEvent, numEvents;
    int32_t refEnt;
    uint32_t slotBase;
    int64_t refIndex;
};
struct MafRenderEvent { // RunMachine::EventLog with the rows' place in the batch's packed rows
    int64_t k;
    uint32_t rowsOff, firstIdx, nRows, _pad;
};
struct MafRenderRow { // RunMachine::PRow
    int64_t key;
    int32_t rank;
    uint32_t ord;
};
struct MafRenderRank { // RunMachine::RankInfo: the sequence of a rank, and the two pieces of text every row of it carries
    int64_t seqStart, srcLength;
    int32_t genome;
    uint32_t headOff, headLen, tailOff, tailLen, _pad;
};
struct MafRenderInput {
    const MafRenderBlock *blocks = nullptr;
    size_t numBlocks = 0;
    const int32_t *entRank = nullptr;
    size_t numEntRank = 0;
    const MafRenderEvent *events = nullptr;
    size_t numEvents = 0;
    const uint32_t *rowEnt = nullptr;
    size_t numRowEnt = 0;
    const MafRenderRow *rows = nullptr;
    size_t numRows = 0;
    const MafRenderRank *ranks = nullptr;
    size_t numRanks = 0;
    const char *chars = nullptr;
    size_t numChars = 0;
    size_t slots = 0; // the blocks' entries, counted block by block (MafRenderBlock::slotBase)
    bool keepEmptyRefBlocks = false;
};
bool mafRenderDevice(hgx_alignment *h, const MafRenderInput &in, char *&text, size_t &bytes);

void columnsReleaseCached(); // idle page-locked memory of the column engine back to the system (hgx_release_cached)

// the per-base tracks kept with the handle for hal2maf (hgx_columns.hip: MafTracks), as a JSON object; and letting go of them
std::string mafTracksInfo(hgx_alignment *h);
void mafTracksDrop(hgx_alignment *h);

} // namespace hgx
