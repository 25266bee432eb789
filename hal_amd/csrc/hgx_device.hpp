// Device-resident alignment tables (HBM layout) and the handle behind the C ABI.
//
// Per genome the image is narrowed to the fields one hop of the segment graph touches together, so that
// a random hop costs one 16-byte record (one 64-byte sector) instead of one sector per field:
//   TopRec<C>[numTop+1]   { C start; i32 parentEnc; i32 paralogy; i32 botParse }   (sentinel: start = length)
//   BotRec<C>[numBot+1]   { C start; i32 topParse }
//   childEnc[slot][numBot] i32
// link encoding: (index << 1) | reversedBit, NULL_INDEX -> -1.  C is int32 when every genome is shorter
// than 2^31 bases, else int64 (tables are re-instantiated, kernels are templates on C).
// Source layouts: api/mmap_impl/mmapTopSegmentData.h:40-44, mmapBottomSegmentData.h:35-52.
#pragma once
#include <mutex>
#include "../../include/hgx.h"
#include "hgx_image.hpp"
#include <array>
#include <map>
#include <memory>

namespace hgx {

template <typename C> struct alignas(16) TopRec;
template <> struct alignas(16) TopRec<int32_t> {
    int32_t start;
    int32_t parentEnc;
    int32_t paralogy;
    int32_t botParse;
};
template <> struct alignas(16) TopRec<int64_t> {
    int64_t start;
    int32_t parentEnc;
    int32_t paralogy;
    int32_t botParse;
    int32_t _pad[3];
};
// Up-walk record of a top segment: what the next level of the walk needs from the PARENT's bottom segment (its start
// and its top-parse index) is copied next to the child's own start and link, so going up one level never touches the
// parent's bottom table.
template <typename C> struct UpRec;
template <> struct alignas(16) UpRec<int32_t> {
    int32_t start;
    int32_t parentEnc;
    int32_t parentStart;    // start coordinate of the parent bottom segment (undefined when parentEnc < 0)
    int32_t parentTopParse; // its topParseIndex (-1 when the parent genome has no top tiling)
};
template <> struct alignas(16) UpRec<int64_t> {
    int64_t start;
    int64_t parentStart;
    int32_t parentEnc;
    int32_t parentTopParse;
    int64_t _pad;
};
// Record of the chained up kernel (k_up_chain): everything one step over one top segment needs in ONE 16-byte load
// (the gather rate of that kernel is bound by the number of L1 misses in flight per CU, so a step must not cost two).
// The segment's length replaces the look at the next record's start.  Two tables per genome: "mid" for a hop whose
// parent is walked further (link = the parent bottom segment's top-parse index) and "last" for the hop into the
// MRCA (link = the parent bottom segment's index).
#ifdef __HIPCC__
#define HGX_HD __host__ __device__
#else
#define HGX_HD // this header is also read by the g++-compiled host files
#endif
template <typename C> struct ChainRec;
template <> struct alignas(16) ChainRec<int32_t> {
    int32_t start;
    int32_t parentStart;  // start coordinate of the parent bottom segment
    uint32_t lenHas;      // (length << 1) | hasParent
    uint32_t linkRev;     // (link << 1) | parentReversed
    HGX_HD int64_t len() const { return (int64_t)(lenHas >> 1); }
    HGX_HD bool hasParent() const { return lenHas & 1u; }
    HGX_HD void set(int64_t s, int64_t ps, int64_t l, bool has, int64_t link, bool rev) {
        start = (int32_t)s;
        parentStart = (int32_t)ps;
        lenHas = ((uint32_t)l << 1) | (has ? 1u : 0u);
        linkRev = ((uint32_t)link << 1) | (rev ? 1u : 0u);
    }
};
template <> struct alignas(16) ChainRec<int64_t> {
    int64_t start;
    int64_t parentStart;
    int64_t length;
    uint32_t linkRev;
    uint32_t has;
    HGX_HD int64_t len() const { return length; }
    HGX_HD bool hasParent() const { return has != 0; }
    HGX_HD void set(int64_t s, int64_t ps, int64_t l, bool h, int64_t link, bool rev) {
        start = s;
        parentStart = ps;
        length = l;
        has = h ? 1u : 0u;
        linkRev = ((uint32_t)link << 1) | (rev ? 1u : 0u);
    }
};
// Record of the down hop (k_down_ring), one per bottom segment and child slot: the child link together with what the hop
// needs from the child's top segment (its start, its length, its paralogy link), so that the hop costs one 16-byte
// gather instead of a gather of the link followed by a gather of the child's record (and of its right neighbour for a
// reversed piece).  Built on first use by a plan (ensureDownTables).
template <typename C> struct DownRec;
template <> struct alignas(16) DownRec<int32_t> {
    int32_t childEnc;   // (child top index << 1) | reversed, -1 = no child
    int32_t childStart; // start coordinate of the child's top segment
    int32_t len;        // its length (= the bottom segment's)
    int32_t paralogy;   // its next paralogy index, -1 = none
};
template <> struct alignas(16) DownRec<int64_t> {
    int64_t childStart;
    int64_t len;
    int32_t childEnc;
    int32_t paralogy;
    int64_t _pad;
};
// Composed up table of a (source genome, ancestor) pair: the pieces the whole up walk (k_up_chain) produces for every
// source top segment, in source order — what a chain file is to a pairwise alignment.  A batch's intervals are then clipped
// against these records instead of walking level by level (k_locate_composed).  Built on the GPU by the walk kernels
// themselves when a plan is created for a large batch (ensureComposedUp).
template <typename C> struct ComposedRec;
template <> struct alignas(16) ComposedRec<int32_t> { // 16 bytes: one gather per record (the kernel is bound by gathers in flight)
    int32_t sLo;    // source genome position of the piece's first base (the source side runs forward)
    int32_t len;
    int32_t so;     // offset of the piece inside the ancestor's bottom segment, in iteration order
    uint32_t mEncF; // (ancestor bottom segment index << 2) | (first piece of its source segment << 1) | target-reversed
};
template <> struct alignas(16) ComposedRec<int64_t> {
    int64_t sLo, so, len;
    uint32_t mEncF;
    uint32_t _pad;
};
struct ComposedUp {
    void *recs = nullptr;       // ComposedRec<C>[numRecs], sorted by sLo
    void *eo = nullptr;         // C[numRecs]: bases of the ancestor's bottom segment after the piece (needed by '-' intervals only)
    uint32_t *coarse = nullptr; // [buckets + 1]: first record that does not end before position bucket << shift
    uint32_t *starts = nullptr; // [buckets + 1] (whole-path tables): first record that begins at or after bucket << shift
    int shift = 0;
    uint64_t numRecs = 0;
    double buildMs = 0;
    bool through = false;       // records hold FINAL pieces in the target genome (so = forward target start, no segment index):
                                // the table composes the whole path source -> MRCA -> target, paralogy rings included
    // The merged form of a whole-path table (hgx_merged_kernels.hpp; 32-bit coordinates only): maximal chains of pieces that
    // canMergeRightWith joins, in (source start, target start) order, with bucket tables that carry the number of flagged
    // records before each entry.  Read by the single-pass kernels of hgx_lift_kernels.hpp.
    void *mRecs = nullptr;      // ComposedRec<int32_t>[mNum]: sLo, len, so = forward target start, mEncF = target strand | sequence << 8
    void *mBuckets = nullptr;   // uint32[buckets + 1]: the first merged record that touches the bucket
    void *mFlagBits = nullptr;  // a bit per bucket: a flagged record touches it (k_lift_general_list)
    int mShift = 0;
    uint64_t mNum = 0, mFlagged = 0;
    int64_t mWindow = 0;        // intervals longer than this take the general path
    double mBuildMs = 0;
    bool mTried = false;        // the merged form was attempted (it stays absent for tables it cannot hold)
};
template <typename C> struct BotRec;
template <> struct alignas(8) BotRec<int32_t> {
    int32_t start;
    int32_t topParse;
};
template <> struct alignas(16) BotRec<int64_t> {
    int64_t start;
    int32_t topParse;
    int32_t _pad;
};

struct DeviceGenome {
    void *top = nullptr;                // TopRec<C>[numTop+1]
    void *up = nullptr;                 // UpRec<C>[numTop+1] (genomes with a parent)
    void *chainMid = nullptr;           // ChainRec<C>[numTop], built on first use by a plan (ensureChainTables)
    void *chainLast = nullptr;          // ChainRec<C>[numTop]
    void *bot = nullptr;                // BotRec<C>[numBot+1]
    std::vector<int32_t *> childEnc;    // per child slot, int32[numBot]
    std::vector<void *> downRec;        // per child slot, DownRec<C>[numBot], built on first use by a plan
    int32_t *locate[2] = {nullptr, nullptr}; // coarse position -> segment index table of the {top, bottom} tiling (ensureLocateTable)
    int locateShift[2] = {0, 0};
    int64_t *seqStart = nullptr;        // int64[numSeq+1] (sentinel = genome length)
    int32_t numSeq = 0;
    int64_t numTop = 0, numBot = 0;
};

// Per-genome descriptor readable from device code (the column kernels walk the whole tree, so they need
// every genome's tables, not just the two or three a liftover launch is given).
struct GenomeDesc {
    const void *top;      // TopRec<C>[numTop+1]
    const void *bot;      // BotRec<C>[numBot+1]
    const int32_t *const *child; // [numChildren] child link arrays (a slice of DeviceImage::childPtrs: no limit on the child count)
    const int32_t *childGenome;  // [numChildren] genome ids of the children (a slice of DeviceImage::childGenomes)
    const uint8_t *dna;   // nibble-packed bases, or null when the alignment carries no DNA
    const int64_t *seqStart;
    int64_t numTop, numBot, length;
    int32_t parent, slotInParent, numChildren, numSeq;
};

struct DeviceImage {
    int device = -1;
    bool wide = false; // C == int64_t
    std::vector<DeviceGenome> genomes;
    GenomeDesc *desc = nullptr;          // device array, one per genome
    const int32_t **childPtrs = nullptr; // device: every genome's child link arrays, one after the other
    int32_t *childGenomes = nullptr;     // device: the children's genome ids, same order
    std::vector<uint8_t *> dna;          // device copies of the packed DNA (uploaded on first use)
    // composed tables: (source genome, ancestor, -1, -1) -> up table; (source, target, dupes, coalescence limit + 1) with
    // through = true -> table of the whole path
    std::map<std::array<int, 4>, ComposedUp> composed;
    size_t bytes = 0;
    ~DeviceImage();
};

// Device workspaces come from a per-device cache of released blocks: hipFree costs a device synchronisation and an unmap
// (releasing the table builder's workspaces was 3.7 of the 8 ms a fresh plan's first batch took, profiles/r03a_bench.log), and the next
// plan asks for the same sizes again.  Sizes are rounded up to a power of two or one and a half times one, so a released block
// fits the next request of its class; blocks beyond HGX_CACHE_BYTES (default 24 GB per device) go back to the driver, and so does
// the whole cache of a device when an alignment on it is closed.  Callers release a block only when no queued work uses it any
// more (the places that used to call hipFree: after a run's synchronisation, when a plan is destroyed).
void *devAlloc(size_t bytes);   // on the current device; throws std::runtime_error
void devRelease(void *p);       // back to the cache of the device it came from (null: nothing)
void devCacheTrim(int device);  // hipFree everything cached for `device`

// builds the k_up_chain tables of `genome` (idempotent, serialised by an internal mutex)
void ensureChainTables(const Image &img, DeviceImage &D, int genome, bool mid, bool last);
// builds the coarse locate table of a genome's top (which = 0) or bottom (1) tiling (idempotent, serialised)
void ensureLocateTable(const Image &img, DeviceImage &D, int genome, int which);
// builds the k_down_ring table of (parent genome, child slot) (idempotent, serialised)
void ensureDownTable(const Image &img, DeviceImage &D, int parent, int slot);

// uploads the packed DNA of every genome (idempotent); needed only by the MAF path
void ensureDeviceDna(const Image &img, DeviceImage &D);

// uploads img to `device`; throws std::runtime_error on any HIP failure or unsupported size
std::unique_ptr<DeviceImage> uploadImage(const Image &img, int device);

} // namespace hgx

// the opaque handle of include/hgx.h
struct hgx_liftover_plan;
struct hgx_alignment {
    // the host image is shared by the handles hgx_clone_to_device makes of one alignment (one copy in host memory, one set of
    // tables per device); `img` is the name every user of a handle knows it by
    std::shared_ptr<hgx::Image> imgHolder;
    hgx::Image &img;
    hgx_alignment() : imgHolder(std::make_shared<hgx::Image>()), img(*imgHolder) {}
    explicit hgx_alignment(const std::shared_ptr<hgx::Image> &shared) : imgHolder(shared), img(*imgHolder) {}
    std::unique_ptr<hgx::DeviceImage> dev;
    // the host-buffer entry points (hgx_liftover_batch, hgx_liftover_convert) keep their last plan: creating one means
    // gigabytes of device allocations, and a BED file is lifted in several batches with the same genomes and options
    struct CachedPlan {
        int src = -1, tgt = -1;
        hgx_liftover_opts opts{};
        size_t maxQueries = 0;
        hgx_liftover_plan *plan = nullptr;
    } cachedPlan;
    // halGetBlocksInTargetRange maps a range forward and the stretches next to its members back: two plans that a browser's
    // calls alternate between, kept like the one above (creating a plan is a millisecond, a call's device work a tenth of it)
    CachedPlan vizPlans[2];
    std::mutex planMutex;
    // hal2maf's per-base tracks of the last (reference, scope, filters) exported from this handle (hgx_columns.hip: MafTracks): the
    // sweeps behind them cover whole genomes, and an export comes in chunks (and, sliced, in many exports)
    std::shared_ptr<void> mafTracks;
    std::mutex mafTracksMutex;
    // pinned host staging of the text path (Liftover::convert): genome coordinates and strands of a batch on the way in, its
    // records on the way out, and the device copy of the former; grown on demand, freed with the alignment
    struct Stage {
        int64_t *gs = nullptr, *ge = nullptr;
        uint8_t *st = nullptr;
        size_t capQ = 0;
        hgx_record *recs = nullptr;
        size_t capR = 0;
        void *dS = nullptr, *dE = nullptr, *dT = nullptr;
        size_t capD = 0;
        // (the 8-byte form of the records for a caller that prints BED lines: device words, pinned words and offsets, a flag)
        void *dPacked = nullptr, *dFlag = nullptr;
        uint32_t *packed = nullptr, *first = nullptr;
        unsigned int *flagHost = nullptr;
        size_t capP = 0, capF = 0;
    } stage;
    ~hgx_alignment();
};
