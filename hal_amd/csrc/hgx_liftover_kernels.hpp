// Device kernels of the liftover walk (gfx950).  Included only by hgx_liftover.hip.
//
// The reference maps one BED interval with a recursive, heap-allocating walk
// (api/impl/halSegmentMapper.cpp:25-330).  Here a batch of intervals is advanced level-synchronously:
// the frontier is a Structure-of-Arrays of "pieces" (query id, source start + strand, current segment
// index, offset inside it, length, strand) in HBM; one kernel per tree hop reads the 16-byte link record
// of each piece's segment, splits pieces where the top and bottom tilings disagree, and appends the
// results to the next frontier with one atomic per wavefront (ballot + popcount prefix).  Integer only.
#pragma once
#include "hgx_device.hpp"
#include "hgx_scan_kernels.hpp"
#include <hip/hip_runtime.h>

namespace hgx {

// frontier / piece flags
static constexpr uint8_t F_SREV = 1; // source iterator reversed
static constexpr uint8_t F_TREV = 2; // current (target-side) iterator reversed
static constexpr uint8_t F_DOT = 4;  // input strand was '.'

struct Frontier {
    int32_t *qid;
    int64_t *sPos; // source position of the piece's first base in iteration order
    int32_t *idx;  // segment index in the current genome
    int64_t *so;   // start offset in iteration order (api/inc/halSegmentIterator.h:115); positional pieces: forward start
    int32_t *len;
    uint8_t *flags;
};

// final mapped pieces in forward coordinates (what MappedSegment::lessThan orders by,
// api/impl/halMappedSegment.cpp:36-43,167-206)
// One 32-byte record per piece (a mapped piece has equal source and target length, so the high ends are implied):
// finalize writes them densely, the grouping scatter moves each with two 16-byte stores into one sector.
struct alignas(16) MappedRec {
    int64_t tLo, sLo;
    int32_t len, qid;
    uint32_t flags, _pad;
};
struct Mapped {
    MappedRec *rec;
};

// counters[] slots
enum {
    CNT_OVERFLOW = 0, // set when any append ran past capacity
    CNT_SRC_PIECES = 4,
    CNT_DEFERRED = 5,
    CNT_MAXNEED = 6,
    CNT_BIGFAIL = 7,
    CNT_MAPPED = 3,    // pieces in the final frontier (written by k_finalize)
    NSEG = 64,         // independent append segments per frontier (one counter word each)
    SEG_PITCH = 128,   // words between the counters of two segments: every segment's counters sit in their own 1 KB, so the
                       // appends of one level land on 64 different cache lines (and memory channels).  Atomics on words that
                       // share a line serialise at the memory side: with the 64 counters packed into four lines the up-phase
                       // kernels spent about a third of their time queueing on them
    CNT_FRONT0 = 8,    // + segment*SEG_PITCH + level: pieces appended to that segment of the level's frontier
    MAX_LEVELS = 120,
    CNT_KSTAT0 = CNT_FRONT0 + NSEG * SEG_PITCH, // + 2*launch: {top, bottom} segment records dereferenced by that launch
    MAX_LAUNCHES = 152,
    CNT_SLOTS = CNT_KSTAT0 + 2 * MAX_LAUNCHES,
    // Statistics (dereference counts per launch, source pieces) are summed by every wavefront at the end of a kernel.
    // Atomics on one cache line complete one after the other at the memory side, about 12 ns each: with 8192 wavefronts
    // adding two or three words of one line every kernel carried a tail of 0.2-0.3 ms (k_locate_through ran 0.21 ms with
    // 1024 blocks and 1.5 ms with 16384).  The sums are therefore spread over STAT_LINES copies, 4 KB apart, chosen by the
    // block index; k_fold_stats adds the copies up at the end of the run into the compact slots above.  Device-only words, after the compact block:
    STAT_LINES = 32,
    STAT_PITCH = 512,      // words between two copies
    STAT_SRC_PIECES = 0,   // word inside a copy
    STAT_MAPPED = 1,       // pieces written by k_locate_through
    STAT_LAUNCH0 = 2,      // + 2*launch: {top, bottom}
    CNT_DSTAT0 = CNT_SLOTS,
    CNT_DEV_SLOTS = CNT_SLOTS + STAT_LINES * STAT_PITCH
};
static_assert(STAT_LAUNCH0 + 2 * MAX_LAUNCHES <= STAT_PITCH, "a copy of the statistics holds two words per launch");
// adds a wavefront's count to this block's copy of a statistics word (word0: the word in copy 0)
__device__ __forceinline__ void stat_add(unsigned long long *word0, uint32_t v) {
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v)
        atomicAdd(word0 + (size_t)(blockIdx.x & (STAT_LINES - 1)) * STAT_PITCH, (unsigned long long)v);
}
static_assert(MAX_LEVELS <= SEG_PITCH, "a segment's counter block holds one word per level");

__device__ __forceinline__ int lane_id() {
    return (int)(threadIdx.x & 63);
}

// adds the copies of the statistics words up into the compact slots (one small launch at the end of a run)
static __global__ void k_fold_stats(unsigned long long *counters, int words) {
    const int w = (int)threadIdx.x;
    if (w >= words)
        return;
    unsigned long long sum = 0;
    for (int l = 0; l < STAT_LINES; ++l)
        sum += counters[CNT_DSTAT0 + (size_t)l * STAT_PITCH + w];
    if (w == STAT_SRC_PIECES)
        counters[CNT_SRC_PIECES] = sum;
    else if (w == STAT_MAPPED)
        counters[CNT_MAPPED] += sum;
    else
        counters[CNT_KSTAT0 + (w - STAT_LAUNCH0)] = sum;
}

// LDS-staged append.  Appending straight to the next frontier costs one device-scope atomic on a single
// counter per wavefront per emit round, and that one word saturates near 90 atomics/us on MI355X — the
// first profile of this path was bound by exactly that.  Each wavefront therefore compacts its emitted
// pieces (ballot + popcount prefix) into a private LDS window and only when the window is nearly full
// reserves a contiguous range of the global frontier with ONE atomic and copies the window out with
// coalesced stores.  The frontier itself is split into NSEG independent segments (block b appends to segment
// b mod NSEG, each with its own counter word), so the remaining atomics spread over 64 addresses instead of
// serialising on one.  Order inside a frontier is irrelevant (the result is re-grouped by query later).
static constexpr int STAGE_CAP = 160;   // pieces per wavefront window (4 waves x 160 x 29 B = 18.5 KB per block)
static constexpr int STAGE_FLUSH = STAGE_CAP - 64;

struct StageMem { // one per block, declared __shared__ by the kernel
    int64_t sPos[4 * STAGE_CAP];
    int32_t qid[4 * STAGE_CAP];
    int32_t idx[4 * STAGE_CAP];
    int64_t so[4 * STAGE_CAP];
    int32_t len[4 * STAGE_CAP];
    uint8_t fl[4 * STAGE_CAP];
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Stage {
    StageMem *m;
    int base;  // this wave's window inside the block's StageMem
    int count; // wave-uniform
    Frontier out;
    unsigned long long *outCount; // this block's segment counter
    unsigned long long *counters;
    uint32_t segCap;              // capacity of one segment
    unsigned long long segBase;   // first slot of this block's segment
    uint32_t *perQuery = nullptr; // final pieces: per-interval count of the pieces actually stored (an overflowing append drops
                                  // pieces; counting at emit time would leave offsets pointing past the buffers)

    // oc: segment 0's counter of the output frontier (segment s at oc + s*SEG_PITCH); cp: total capacity of the frontier buffers
    __device__ __forceinline__ void init(StageMem *mem, const Frontier &o, unsigned long long *oc, unsigned long long *c, uint32_t cp) {
        m = mem;
        base = (int)(threadIdx.x >> 6) * STAGE_CAP;
        count = 0;
        out = o;
        const uint32_t seg = blockIdx.x % NSEG;
        outCount = oc + (size_t)seg * SEG_PITCH;
        counters = c;
        segCap = cp / NSEG;
        segBase = (unsigned long long)seg * segCap;
    }
    __device__ __forceinline__ void flush() {
        wave_lds_fence();
        const int n = count;
        if (n > 0) {
            const int lane = lane_id();
            unsigned long long b = 0;
            if (lane == 0)
                b = atomicAdd(outCount, (unsigned long long)n);
            b = __shfl(b, 0);
            for (int i = lane; i < n; i += 64) {
                const unsigned long long off = b + (unsigned long long)i;
                const unsigned long long slot = segBase + off;
                if (off < segCap) {
                    out.qid[slot] = m->qid[base + i];
                    out.sPos[slot] = m->sPos[base + i];
                    out.idx[slot] = m->idx[base + i];
                    out.so[slot] = m->so[base + i];
                    out.len[slot] = m->len[base + i];
                    out.flags[slot] = m->fl[base + i];
                    if (perQuery)
                        atomicAdd(&perQuery[m->qid[base + i]], 1u);
                } else {
                    counters[CNT_OVERFLOW] = 1;
                }
            }
            count = 0;
        }
        wave_lds_fence();
    }
    // all lanes of the wave call this together; lanes with emit == true contribute one piece
    __device__ __forceinline__ void emit(bool doEmit, int32_t qid, int64_t sPos, int32_t idx, int64_t so, int32_t len, uint8_t fl) {
        const unsigned long long mask = __ballot(doEmit);
        if (mask == 0)
            return;
        if (doEmit) {
            const int p = base + count + (int)__popcll(mask & ((1ull << lane_id()) - 1ull));
            m->qid[p] = qid;
            m->sPos[p] = sPos;
            m->idx[p] = idx;
            m->so[p] = so;
            m->len[p] = len;
            m->fl[p] = fl;
        }
        count += (int)__popcll(mask);
        if (count > STAGE_FLUSH)
            flush();
    }
};


// Reading a segmented frontier: linear work index -> physical slot.  Each block scans the NSEG segment counts
// once into LDS; a work item finds its segment by binary search over the 65 prefix sums.
struct FrontView {
    uint32_t prefix[NSEG + 1];
};
__device__ __forceinline__ uint32_t front_view_init(FrontView *v, const unsigned long long *segCount, uint32_t cap) {
    const uint32_t segCap = cap / NSEG;
    constexpr int PER = NSEG / 64; // the first wavefront scans the counts, PER consecutive segments per lane
    if (threadIdx.x < 64) {
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const unsigned long long x = segCount[(size_t)(threadIdx.x * PER + k) * SEG_PITCH];
            c[k] = (uint32_t)(x < segCap ? x : segCap);
            sum += c[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o)
                incl += up;
        }
        uint32_t acc = incl - sum;
        if (threadIdx.x == 0)
            v->prefix[0] = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            acc += c[k];
            v->prefix[threadIdx.x * PER + k + 1] = acc;
        }
    }
    __syncthreads();
    return v->prefix[NSEG];
}
__device__ __forceinline__ uint32_t front_slot(const FrontView *v, uint32_t i, uint32_t cap) {
    int lo = 0, hi = NSEG; // prefix[lo] <= i < prefix[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (v->prefix[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    return (uint32_t)lo * (cap / NSEG) + (i - v->prefix[lo]);
}

// ---------------------------------------------------------------------------------------------
// Stage 0: locate + expand.  BlockLiftover::liftInterval, liftover/impl/halBlockLiftover.cpp:46-72:
// toSite(globalStart) (api/impl/halSegmentIterator.cpp:240-299, here a binary search on start[]), slice,
// then one halMapSegment call per source segment until globalEnd; '-' intervals flip each piece in place
// (:64-70, toReverseInPlace = flip strand and swap offsets, halSegmentIterator.cpp:149-153).
// One lane per interval; the lanes of a wave walk their source segments in lockstep and append.
template <typename REC>
__global__ void __launch_bounds__(256) k_locate_expand(const REC *__restrict__ segs, int64_t numSegs, const int64_t *__restrict__ gStart,
                                                       const int64_t *__restrict__ gEnd, const uint8_t *__restrict__ strand,
                                                       uint32_t nq, const int32_t *__restrict__ coarse, int coarseShift, Frontier out,
                                                       uint32_t cap,
                                                       unsigned long long *counters, unsigned long long *kstat) {
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t derefs = 0;
    __shared__ StageMem stageMem;
    Stage stage;
    stage.init(&stageMem, out, &counters[CNT_FRONT0], counters, cap);
    for (uint32_t base = wave * 64; base < nq; base += wavesTotal * 64) {
        const uint32_t q = base + lane_id();
        bool act = q < nq;
        int64_t gs = 0, ge = -1;
        uint8_t fl = 0;
        bool minus = false;
        int64_t j = 0;
        if (act) {
            gs = gStart[q];
            ge = gEnd[q];
            const uint8_t st = strand[q];
            minus = st == '-';
            if (st == '.')
                fl |= F_DOT;
            act = ge >= gs && gs >= 0 && numSegs > 0 && gs < (int64_t)segs[numSegs].start;
            // largest j with start[j] <= gs: toSite (halSegmentIterator.cpp:240-299; interpolation search there).  The coarse
            // table (segment index at every 2^shift-th position, ~4 segments per bucket) narrows the binary search to a few
            // probes on neighbouring records instead of ~20 probes all over a 16 MB table
            int64_t lo = 0, hi = numSegs; // invariant start[lo] <= gs < start[hi]
            if (act && coarse) {
                const int64_t b = gs >> coarseShift;
                lo = coarse[b];
                const int64_t up = (int64_t)coarse[b + 1] + 1;
                hi = up < numSegs ? up : numSegs;
                if (hi <= lo)
                    hi = lo + 1;
            }
            if (act) {
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if ((int64_t)segs[mid].start <= gs)
                        lo = mid;
                    else
                        hi = mid;
                }
            }
            j = lo;
        }
        int64_t curStart = act ? (int64_t)segs[j].start : 0;
        while (__any(act)) {
            int64_t nextStart = 0;
            bool emit = false;
            int64_t plo = 0, phi = 0;
            if (act) {
                nextStart = (int64_t)segs[j + 1].start;
                ++derefs;
                plo = gs > curStart ? gs : curStart;
                phi = ge < nextStart - 1 ? ge : nextStart - 1;
                emit = true;
            }
            // one converged call: emit() ballots the whole wavefront
            stage.emit(emit, (int32_t)q, minus ? phi : plo, (int32_t)j, (int32_t)(minus ? nextStart - 1 - phi : plo - curStart),
                       (int32_t)(phi - plo + 1), minus ? (uint8_t)(fl | F_SREV | F_TREV) : fl);
            if (emit) {
                ++j;
                curStart = nextStart;
                act = j < numSegs && curStart <= ge;
            }
        }
    }
    stage.flush();
    stat_add(&counters[CNT_DSTAT0 + STAT_SRC_PIECES], derefs);
    stat_add(kstat, derefs);
}

// ---------------------------------------------------------------------------------------------
// Stage 0 + the whole up phase from the composed table (ComposedRec, hgx_device.hpp): the records of consecutive source
// segments are contiguous, so an interval is one binary search, one look at pstart[] and a scan over the few records it
// overlaps; each is clipped to the interval and leaves as an ordinary bottom piece in the ancestor — the same pieces
// k_locate_expand + k_up_chain produce (a source segment's pieces are cut at the same places whatever part of the segment
// an interval covers, so clipping the whole segment's pieces is the same as walking the part).  A '-' interval is the
// forward result reversed in place: source first base = the high end, both strand bits flipped, offsets from the other end
// (toReverseInPlace commutes with every hop: parent / child hops copy the offsets, parse steps are symmetric).
// The kernel is bound, like the walk kernels, by the number of gathers a CU keeps in flight (PMC: 21.6 M L1 requests per
// 1 M intervals with 32-byte records found through a binary search on the source tiling and a per-segment index), so a
// record is 16 bytes and one gather, the bases after the piece (needed to reverse it) live in a side array read by '-'
// intervals only, and a coarse table over source positions leads straight to the first record that can overlap.
// Eight lanes work on one interval: they load eight consecutive records (one 128-byte line) per round, clip them in
// parallel and emit together, so an interval with twenty pieces takes three rounds instead of twenty dependent gathers in
// one lane while its neighbours wait (one lane per interval ran at 0.43-0.46 ms, this at a third of it).
template <typename C>
__global__ void __launch_bounds__(256) k_locate_composed(const int64_t *__restrict__ gStart, const int64_t *__restrict__ gEnd,
                                                         const uint8_t *__restrict__ strand, uint32_t nq, int64_t genomeLength,
                                                         const uint32_t *__restrict__ coarse, int coarseShift,
                                                         const ComposedRec<C> *__restrict__ recs, const C *__restrict__ eoOf, uint64_t numRecs,
                                                         Frontier out, uint32_t cap, unsigned long long *outCount,
                                                         unsigned long long *counters, unsigned long long *kstat) {
    constexpr int G = 4, PER_WAVE = 64 / G;
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id(), li = lane & (G - 1);
    uint32_t srcPieces = 0, used = 0; // used: table records that overlap their interval (one piece each)
    __shared__ StageMem stageMem;
    Stage stage;
    stage.init(&stageMem, out, outCount, counters, cap);
    for (uint32_t base = wave * PER_WAVE; base < nq; base += wavesTotal * PER_WAVE) {
        const uint32_t q = base + (uint32_t)(lane / G);
        bool act = q < nq; // the same in all lanes of a group
        int64_t gs = 0, ge = -1;
        uint8_t dot = 0;
        bool minus = false, first = true;
        uint64_t k = 0;
        if (act) {
            gs = gStart[q];
            ge = gEnd[q];
            const uint8_t st = strand[q];
            minus = st == '-';
            if (st == '.')
                dot = F_DOT;
            act = ge >= gs && gs >= 0 && gs < genomeLength && numRecs > 0;
            if (act)
                k = coarse[gs >> coarseShift];
        }
        while (__any(act)) {
            bool emit = false, beyond = false, startsSegment = false;
            int32_t oIdx = 0, oLen = 0;
            int64_t oSPos = 0, oSo = 0;
            uint8_t oFl = 0;
            if (act) {
                const uint64_t at = k + (uint64_t)li;
                if (at >= numRecs) {
                    beyond = true;
                } else {
                    const ComposedRec<C> r = recs[at];
                    const int64_t pLo = (int64_t)r.sLo, pHi = pLo + (int64_t)r.len - 1;
                    if (pLo > ge) {
                        beyond = true;
                    } else if (pHi >= gs) {
                        const int64_t c = pLo > gs ? pLo : gs, d = pHi < ge ? pHi : ge;
                        const int64_t n = d - c + 1, delta = c - pLo;
                        emit = true;
                        oLen = (int32_t)n;
                        oIdx = (int32_t)(r.mEncF >> 2);
                        oFl = (uint8_t)(((r.mEncF & 1u) ? F_TREV : 0) | dot);
                        if (!minus) {
                            oSPos = c;
                            oSo = (int64_t)r.so + delta;
                        } else {
                            oSPos = d;
                            oSo = (int64_t)eoOf[at] + ((int64_t)r.len - delta - n);
                            oFl ^= (uint8_t)(F_SREV | F_TREV);
                        }
                        startsSegment = (r.mEncF & 2u) != 0;
                    }
                }
            }
            // the interval is finished once any of its lanes saw a record beyond its end (records are in source order)
            const unsigned long long bmask = __ballot(beyond);
            const unsigned long long gmask = ((1ull << G) - 1ull) << (lane & ~(G - 1));
            // source segments that contribute (the statistics' "source pieces"): every emitted record that starts its segment,
            // and the interval's very first emitted record if it does not
            const unsigned long long emask = __ballot(emit) & gmask;
            if (emit && (startsSegment || (first && (emask & ((1ull << lane) - 1ull)) == 0)))
                ++srcPieces;
            if (emask)
                first = false;
            used += emit ? 1u : 0u;
            stage.emit(emit, (int32_t)q, oSPos, oIdx, oSo, oLen, oFl);
            if (act) {
                if (bmask & gmask)
                    act = false;
                else
                    k += G;
            }
        }
    }
    stage.flush();
    stat_add(&counters[CNT_DSTAT0 + STAT_SRC_PIECES], srcPieces);
    stat_add(&kstat[0], used); // reported in the "top" slot of this launch: records of the composed table dereferenced
}

// ---------------------------------------------------------------------------------------------
// The whole path from one table.  The records are the FINAL pieces of every source top segment (source -> MRCA -> target
// with the paralogy rings and, if asked for, the coalescenceLimit phase; built by running each segment through the walk
// kernels once, buildComposed): (source start, length, forward target start, target strand).  Records of one source segment
// may overlap each other in the source (paralogs), so the coarse table gives the first record that can overlap a position
// and `starts` the first record that begins at or after it; an interval's pieces are among records
// [coarse[start bucket], starts[end bucket + 1]) — an upper bound on its piece count known before any record is read.
// That bound lets the kernel write the pieces grouped by interval straight away: a wavefront (16 intervals, four lanes
// each) reserves the sum of its bounds with one atomic on its block's segment counter and every interval writes its
// MappedRecs contiguously from its own offset.  offset[] and perQuery[] are then exactly what the finishing kernels read —
// no frontier, no per-interval atomics, no scan and no grouping scatter; the price is holes in the piece buffer (about
// as many slots as pieces), which nothing reads.
template <typename C>
__global__ void __launch_bounds__(256) k_locate_through(const int64_t *__restrict__ gStart, const int64_t *__restrict__ gEnd,
                                                        const uint8_t *__restrict__ strand, uint32_t nq, int64_t genomeLength,
                                                        const uint32_t *__restrict__ coarse, const uint32_t *__restrict__ starts, int coarseShift,
                                                        const ComposedRec<C> *__restrict__ recs, Mapped out, uint32_t cap,
                                                        unsigned long long *segCounters, unsigned long long *counters,
                                                        unsigned long long *kstat, uint32_t *__restrict__ offset,
                                                        uint32_t *__restrict__ perQuery, const uint32_t *__restrict__ qlist = nullptr,
                                                        const unsigned long long *__restrict__ qcount = nullptr) {
    // qlist / qcount: work on these intervals only (the general intervals of the merged-table path, hgx_lift_kernels.hpp)
    if (qlist)
        nq = (uint32_t)*qcount;
    constexpr int G = 4, PER_WAVE = 64 / G;
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id(), li = lane & (G - 1);
    const uint32_t seg = blockIdx.x % NSEG, segCap = cap / NSEG;
    const uint32_t segBase = seg * segCap;
    unsigned long long *segCount = segCounters + (size_t)seg * SEG_PITCH;
    const unsigned long long gmask = ((1ull << G) - 1ull) << (lane & ~(G - 1));
    uint32_t srcPieces = 0, used = 0;
    for (uint32_t base = wave * PER_WAVE; base < nq; base += wavesTotal * PER_WAVE) {
        const uint32_t slot = base + (uint32_t)(lane / G);
        bool act = slot < nq; // the same in all lanes of a group
        const uint32_t q = (qlist && act) ? qlist[slot] : slot;
        int64_t gs = 0, ge = -1;
        uint32_t dot = 0;
        bool minus = false, first = true;
        uint32_t k = 0, kEnd = 0;
        if (act) {
            gs = gStart[q];
            ge = gEnd[q];
            const uint8_t st = strand[q];
            minus = st == '-';
            if (st == '.')
                dot = F_DOT;
            act = ge >= gs && gs >= 0 && gs < genomeLength;
            if (act) {
                const int64_t geIn = ge < genomeLength ? ge : genomeLength - 1;
                k = coarse[gs >> coarseShift];
                kEnd = starts[(geIn >> coarseShift) + 1];
                act = kEnd > k;
            }
        }
        // the first round's records are requested before the reservation, so the gathers do not wait for the atomic
        ComposedRec<C> r{};
        bool inRange = act && k + (uint32_t)li < kEnd;
        if (inRange)
            r = recs[k + (uint32_t)li];
        // reserve room for the wavefront's intervals: prefix sum of the bounds over the groups, one atomic
        const uint32_t bound = act ? kEnd - k : 0u;
        uint32_t incl = li == 0 ? bound : 0u;
        const uint32_t own = incl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o)
                incl += up;
        }
        const uint32_t total = __shfl(incl, 63);
        uint32_t myOff = __shfl(incl - own, lane & ~(G - 1)); // the group leader's exclusive prefix
        if (total) {
            unsigned long long b = 0;
            if (lane == 0)
                b = atomicAdd(segCount, (unsigned long long)total);
            b = __shfl(b, 0);
            if (b + total > segCap) { // the retry sizes the buffer from the counters
                if (lane == 0)
                    counters[CNT_OVERFLOW] = 1;
                act = false;
                k = kEnd;
            }
            myOff += segBase + (uint32_t)b;
        }
        uint32_t written = 0; // pieces of this interval so far (the same in all lanes of a group)
        while (__any(act)) {
            bool emit = false, beyond = false, startsSegment = false;
            MappedRec m;
            if (act) {
                if (!inRange) {
                    beyond = true;
                } else {
                    const int64_t pLo = (int64_t)r.sLo, pHi = pLo + (int64_t)r.len - 1;
                    if (pLo > ge) {
                        beyond = true;
                    } else if (pHi >= gs) {
                        const int64_t c = pLo > gs ? pLo : gs, d = pHi < ge ? pHi : ge;
                        const int64_t n = d - c + 1, delta = c - pLo;
                        emit = true;
                        // forward low ends (what the FINAL down hop emits); a reversed target counts from the other end.
                        // A '-' interval is the same piece with both strand bits flipped.
                        m.sLo = c;
                        m.tLo = (int64_t)r.so + ((r.mEncF & 1u) ? (int64_t)r.len - delta - n : delta);
                        m.len = (int32_t)n;
                        m.qid = (int32_t)q;
                        m.flags = (((r.mEncF & 1u) ? F_TREV : 0u) | dot) ^ (minus ? (uint32_t)(F_SREV | F_TREV) : 0u);
                        m._pad = 0;
                        startsSegment = (r.mEncF & 2u) != 0;
                    }
                }
            }
            const unsigned long long bmask = __ballot(beyond);
            const unsigned long long emask = __ballot(emit) & gmask;
            if (emit) {
                if (startsSegment || (first && (emask & ((1ull << lane) - 1ull)) == 0))
                    ++srcPieces;
                ++used;
                out.rec[myOff + written + (uint32_t)__popcll(emask & ((1ull << lane) - 1ull))] = m;
            }
            if (emask)
                first = false;
            written += (uint32_t)__popcll(emask);
            if (act) {
                if (bmask & gmask) {
                    act = false;
                } else {
                    k += G;
                    inRange = k + (uint32_t)li < kEnd;
                    if (inRange)
                        r = recs[k + (uint32_t)li];
                }
            }
        }
        if (li == 0 && slot < nq) {
            offset[q] = written ? myOff : 0u;
            perQuery[q] = written;
        }
    }
    stat_add(&counters[CNT_DSTAT0 + STAT_SRC_PIECES], srcPieces);
    stat_add(&counters[CNT_DSTAT0 + STAT_MAPPED], used);
    stat_add(&kstat[0], used); // reported in the "top" slot of this launch: records of the composed table dereferenced
}

// ---------------------------------------------------------------------------------------------
// Up phase.  mapUp (halSegmentMapper.cpp:25-80) alternates two steps per level: a top piece goes to its parent's
// bottom segment (toParent, api/impl/halBottomSegmentIterator.cpp:40-49: index = parentIndex, offsets copied,
// strand ^= parentReversed; dropped without a parent or below minLength; doDupes is always true on the way up, :108),
// and the resulting bottom piece is re-expressed on the parent genome's top tiling (toParseUp,
// api/impl/halTopSegmentIterator.cpp:55-81, + the toRight(rightCutoff) loop of :46-77), which splits it wherever the
// two tilings disagree, the source being sliced by the same deltas (:52-62).
// With UpRec the parent's start and top-parse index come with the child's record, so a piece travels between levels
// as "forward start in the parent genome + length + top-parse hint" and one level costs only the walk over the
// parent genome's UpRec table.  The hop into the MRCA emits ordinary bottom pieces (index + offset) instead.

// source top pieces -> parent (positional, or ordinary when the parent is the MRCA)
template <typename C>
__global__ void __launch_bounds__(256) k_up_first(const UpRec<C> *__restrict__ up, Frontier in, const unsigned long long *inCount, uint32_t cap,
                                                  Frontier out, unsigned long long *outCount, int64_t minLength, int last,
                                                  unsigned long long *counters, unsigned long long *kstat) {
    __shared__ FrontView fview;
    const uint32_t n = front_view_init(&fview, inCount, cap);
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t topDerefs = 0, botDerefs = 0;
    __shared__ StageMem stageMem;
    Stage stage;
    stage.init(&stageMem, out, outCount, counters, cap);
    for (uint32_t base = wave * 64; base < n; base += wavesTotal * 64) {
        const uint32_t li = base + lane_id();
        const uint32_t i = li < n ? front_slot(&fview, li, cap) : 0;
        bool emit = false;
        int32_t qid = 0, oIdx = 0, len = 0;
        int64_t sPos = 0, oSo = 0;
        uint8_t fl = 0;
        if (li < n) {
            const int32_t t = in.idx[i];
            len = in.len[i];
            const UpRec<C> r = up[t];
            ++topDerefs;
            if (r.parentEnc >= 0 && (int64_t)len >= minLength) {
                emit = true;
                qid = in.qid[i];
                sPos = in.sPos[i];
                const int32_t so = (int32_t)in.so[i];
                fl = in.flags[i];
                if (r.parentEnc & 1)
                    fl ^= F_TREV;
                if (last) {
                    oIdx = r.parentEnc >> 1;
                    oSo = so;
                } else {
                    const int64_t segLen = (int64_t)up[t + 1].start - (int64_t)r.start;
                    oIdx = r.parentTopParse;
                    oSo = !(fl & F_TREV) ? (int64_t)r.parentStart + so : (int64_t)r.parentStart + segLen - so - len;
                }
            }
        }
        stage.emit(emit, qid, sPos, oIdx, oSo, len, fl);
    }
    stage.flush();
    stat_add(&kstat[0], topDerefs);
    stat_add(&kstat[1], botDerefs);
}

// positional pieces in genome P -> split on P's top tiling -> P's parent (positional, or ordinary when that is the MRCA)
template <typename C>
__global__ void __launch_bounds__(256) k_up_walk(const UpRec<C> *__restrict__ up, Frontier in, const unsigned long long *inCount, uint32_t cap,
                                                 Frontier out, unsigned long long *outCount, int64_t minLength, int last,
                                                 unsigned long long *counters, unsigned long long *kstat) {
    __shared__ FrontView fview;
    const uint32_t n = front_view_init(&fview, inCount, cap);
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t topDerefs = 0, botDerefs = 0;
    __shared__ StageMem stageMem;
    Stage stage;
    stage.init(&stageMem, out, outCount, counters, cap);
    for (uint32_t base = wave * 64; base < n; base += wavesTotal * 64) {
        const uint32_t li = base + lane_id();
        const uint32_t i = li < n ? front_slot(&fview, li, cap) : 0;
        bool act = li < n;
        int32_t qid = 0;
        int64_t sPos = 0, lo = 0, hi = -1;
        uint8_t fl = 0;
        int64_t j = 0;
        UpRec<C> cur;
        cur.start = 0;
        cur.parentEnc = -1;
        cur.parentStart = 0;
        cur.parentTopParse = -1;
        if (act) {
            qid = in.qid[i];
            sPos = in.sPos[i];
            fl = in.flags[i];
            lo = in.so[i];
            hi = lo + in.len[i] - 1;
            j = in.idx[i];
            // toParseUp's "while startPos >= segStart+segLen ++index" (halTopSegmentIterator.cpp:64-66), applied to the
            // piece's left end (the right end is reached by the toRight loop)
            while ((int64_t)up[j + 1].start <= lo) {
                ++j;
                ++topDerefs;
            }
            cur = up[j];
            ++topDerefs;
            ++botDerefs; // logical dereference of the bottom segment this piece came from
        }
        while (__any(act)) {
            bool emit = false;
            int32_t oIdx = 0, oLen = 0;
            int64_t oSPos = 0, oSo = 0;
            uint8_t oFl = fl;
            UpRec<C> nxt = cur;
            if (act) {
                nxt = up[j + 1];
                const int64_t curStart = (int64_t)cur.start, nextStart = (int64_t)nxt.start;
                const int64_t plo = lo > curStart ? lo : curStart;
                const int64_t phi = hi < nextStart - 1 ? hi : nextStart - 1;
                oLen = (int32_t)(phi - plo + 1);
                if (cur.parentEnc >= 0 && (int64_t)oLen >= minLength) {
                    emit = true;
                    int64_t so, d; // offset in the top segment, and distance from the piece's first base (iteration order)
                    if (!(fl & F_TREV)) {
                        so = plo - curStart;
                        d = plo - lo;
                    } else {
                        so = nextStart - 1 - phi;
                        d = hi - phi;
                    }
                    oSPos = (fl & F_SREV) ? sPos - d : sPos + d;
                    if (cur.parentEnc & 1)
                        oFl ^= F_TREV;
                    if (last) {
                        oIdx = cur.parentEnc >> 1;
                        oSo = so;
                    } else {
                        oIdx = cur.parentTopParse;
                        const int64_t segLen = nextStart - curStart;
                        oSo = !(oFl & F_TREV) ? (int64_t)cur.parentStart + so : (int64_t)cur.parentStart + segLen - so - oLen;
                    }
                }
            }
            stage.emit(emit, qid, oSPos, oIdx, oSo, oLen, oFl);
            if (act) {
                ++j;
                cur = nxt;
                act = (int64_t)cur.start <= hi;
                if (act)
                    ++topDerefs;
            }
        }
    }
    stage.flush();
    stat_add(&kstat[0], topDerefs);
    stat_add(&kstat[1], botDerefs);
}

// The whole up phase in one launch.  With k_up_first/k_up_walk every level writes its pieces to HBM and the next launch
// reads them back (29 B each way per piece per level) although a piece's way up depends on nothing but itself.  Here a
// lane takes one source piece and follows it to the MRCA depth first: a step looks at one top segment of the current
// level (one UpRec gather), cuts the piece against it, and either emits the cut (last level), descends with it to the
// next level, or moves right; a level that still has bases to the right is parked on a small per-lane stack.  The
// results are the same set of pieces k_up_walk produces level by level (the order inside a frontier is irrelevant).
// Lanes that finish refill from the wavefront's private slice of the input, so the wave stays full until its slice
// is exhausted.  Dereference counters follow the same rules as k_up_first / k_up_walk.
static constexpr int MAX_CHAIN = 16; // up hops handled by the chained kernel (deeper paths use one launch per level)
template <typename C> struct UpTables {
    const ChainRec<C> *t[MAX_CHAIN]; // t[k] = chain table of the k-th genome on the way up (t[0] = source); "last" flavour for t[n-1]
    int n;                           // number of hops
};

// anchor of the affine map forward position -> source position of a piece whose first base (iteration order) has
// source position sPos: first base is at lo on the forward target strand, at hi on the reverse one
__device__ __forceinline__ int64_t make_anchor(int64_t sPos, int64_t lo, int64_t hi, uint8_t fl) {
    const int64_t first = !(fl & F_TREV) ? lo : hi;
    const bool neg = ((fl & F_SREV) != 0) != ((fl & F_TREV) != 0);
    return neg ? sPos + first : sPos - first;
}
// What bounds this kernel (PMC, profiles/r01i_*): the number of cache-line requests a CU's L1 keeps in flight — about
// 60 at ~630 cycles each, TCP pending-stall ~75 % of the time — so the cost of a step is its number of L1 requests,
// and a step is exactly one 16-byte gather (ChainRec).  Tried and measured slower: a register window of 4-5 records
// per step (more requests per step than scan steps saved), and 4 lanes per piece sharing one 64-byte request (fewer
// requests, but four times the wave-steps made it issue-bound: 3.1 ms against 1.2 ms), and fetching the right-hand
// neighbour together with the record when a piece has just arrived at a level, to save the first scan step (1.23 ms against 1.20 ms).
// Doing the single down hop into the target inside this kernel as well (a lane that has cut a piece at the last level gathers
// the DownRec in its next step and walks the ring) was also measured: 1.68 ms against 1.21 + 0.48 ms for the two launches — the
// frontier between them is coalesced traffic, the gathers are what costs, and fusing does not remove any.
static inline size_t upChainLdsBytes(int hops) {
    const int slots = hops - 1 > 1 ? hops - 2 : 1;
    return (size_t)slots * 256 * (8 + 8 + 4 + 1);
}

template <typename C>
__global__ void __launch_bounds__(256) k_up_chain(UpTables<C> tabs, Frontier in, const unsigned long long *inCount, uint32_t cap,
                                                  Frontier out, unsigned long long *outCount, int64_t minLength,
                                                  unsigned long long *counters, unsigned long long *kstat) {
    __shared__ FrontView fview;
    __shared__ const ChainRec<C> *sTab[MAX_CHAIN];
    if (threadIdx.x < MAX_CHAIN)
        sTab[threadIdx.x] = tabs.t[threadIdx.x];
    const uint32_t n = front_view_init(&fview, inCount, cap); // contains the block barriers that publish sTab
    const int lastLvl = tabs.n - 1;
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t per = (n + wavesTotal - 1) / wavesTotal;
    uint32_t next = wave * per; // wave-uniform cursor into this wave's slice
    const uint32_t endItem = (uint64_t)next + per < n ? next + per : n;
    uint32_t topDerefs = 0, botDerefs = 0;
    __shared__ StageMem stageMem;
    Stage stage;
    stage.init(&stageMem, out, outCount, counters, cap);

    // parked levels live in dynamic LDS, [slot][thread]: slot = level - 1, levels 1 .. lastLvl-1 can park.
    // A parked level keeps what its remaining segments need: the right clip, the next segment, the flags and the
    // affine map from a forward position to the source position (see make_anchor).
    extern __shared__ __align__(16) unsigned char dynLds[];
    const int slots = lastLvl > 1 ? lastLvl - 1 : 1;
    int64_t *stAnchor = (int64_t *)dynLds;
    int64_t *stHi = stAnchor + slots * 256;
    int32_t *stJ = (int32_t *)(stHi + slots * 256);
    uint8_t *stFl = (uint8_t *)(stJ + slots * 256);
    uint32_t pending = 0;

    bool busy = false, fresh = false;
    int lvl = 0;
    int32_t qid = 0;
    int64_t j = 0, lo = 0, hi = -1, anchor = 0;
    uint8_t fl = 0;
    for (;;) {
        // refill idle lanes from the slice: level 0 is the source piece itself (never splits: it lies inside one top
        // segment), mapped to its parent here
        const unsigned long long idle = __ballot(!busy);
        if (idle != 0 && next < endItem) {
            const uint32_t mine = next + (uint32_t)__popcll(idle & ((1ull << lane_id()) - 1ull));
            if (!busy && mine < endItem) {
                const uint32_t i = front_slot(&fview, mine, cap);
                const int32_t t = in.idx[i];
                const int32_t len = in.len[i];
                const ChainRec<C> r = sTab[0][t];
                ++topDerefs;
                if (r.hasParent() && (int64_t)len >= minLength) {
                    qid = in.qid[i];
                    const int64_t so = in.so[i];
                    fl = in.flags[i];
                    if (r.linkRev & 1u)
                        fl ^= F_TREV;
                    lo = !(fl & F_TREV) ? (int64_t)r.parentStart + so : (int64_t)r.parentStart + r.len() - so - len;
                    hi = lo + len - 1;
                    anchor = make_anchor(in.sPos[i], lo, hi, fl);
                    j = (int64_t)(r.linkRev >> 1);
                    lvl = 1;
                    fresh = true;
                    busy = true;
                    pending = 0;
                }
            }
            next += (uint32_t)__popcll(idle);
            if (next > endItem)
                next = endItem;
        }
        if (!__any(busy)) {
            if (next >= endItem)
                break;
            continue;
        }
        bool emit = false;
        int32_t oIdx = 0, oLen = 0;
        int64_t oSPos = 0, oSo = 0;
        uint8_t oFl = 0;
        if (busy) {
            const ChainRec<C> r = sTab[lvl][j]; // the step's only memory access
            const int64_t curStart = (int64_t)r.start, nextStart = curStart + r.len();
            ++topDerefs;
            if (fresh && nextStart <= lo) {
                ++j; // toParseUp's linear scan to the segment holding the piece's left end (halTopSegmentIterator.cpp:64-66)
            } else {
                if (fresh) {
                    fresh = false;
                    ++botDerefs; // logical dereference of the bottom segment this piece came from
                }
                const int64_t plo = lo > curStart ? lo : curStart;
                const int64_t phi = hi < nextStart - 1 ? hi : nextStart - 1;
                oLen = (int32_t)(phi - plo + 1);
                const bool more = nextStart <= hi;
                const bool valid = r.hasParent() && (int64_t)oLen >= minLength;
                ++j;
                bool descended = false;
                if (valid) {
                    // source position of the cut's first base in iteration order: srcPos(p) = anchor +- p
                    const int64_t so = !(fl & F_TREV) ? plo - curStart : nextStart - 1 - phi;
                    const int64_t edge = !(fl & F_TREV) ? plo : phi;
                    const bool neg = ((fl & F_SREV) != 0) != ((fl & F_TREV) != 0);
                    oSPos = neg ? anchor - edge : anchor + edge;
                    oFl = fl;
                    if (r.linkRev & 1u)
                        oFl ^= F_TREV;
                    if (lvl == lastLvl) {
                        emit = true;
                        oIdx = (int32_t)(r.linkRev >> 1);
                        oSo = so;
                    } else {
                        if (more) {
                            const int at = (lvl - 1) * 256 + (int)threadIdx.x;
                            stJ[at] = (int32_t)j;
                            stAnchor[at] = anchor;
                            stHi[at] = hi;
                            stFl[at] = fl;
                            pending |= 1u << lvl;
                        }
                        lo = !(oFl & F_TREV) ? (int64_t)r.parentStart + so : (int64_t)r.parentStart + r.len() - so - oLen;
                        hi = lo + oLen - 1;
                        fl = oFl;
                        anchor = make_anchor(oSPos, lo, hi, fl);
                        j = (int64_t)(r.linkRev >> 1);
                        ++lvl;
                        fresh = true;
                        descended = true;
                    }
                }
                if (!descended && !more) {
                    if (pending) {
                        lvl = 31 - __clz((int)pending);
                        pending &= ~(1u << lvl);
                        const int at = (lvl - 1) * 256 + (int)threadIdx.x;
                        j = stJ[at];
                        anchor = stAnchor[at];
                        hi = stHi[at];
                        fl = stFl[at];
                        lo = INT64_MIN; // a resumed level starts at a segment boundary: plo = curStart
                    } else {
                        busy = false;
                    }
                }
            }
        }
        stage.emit(emit, qid, oSPos, oIdx, oSo, oLen, oFl);
    }
    stage.flush();
    stat_add(&kstat[0], topDerefs);
    stat_add(&kstat[1], botDerefs);
}

// ---------------------------------------------------------------------------------------------
// D (+R): bottom piece -> child's top piece through child slot `slot` (mapDown bottom branch,
// halSegmentMapper.cpp:133-143; TopSegmentIterator::toChild, api/impl/halTopSegmentIterator.cpp:36-45),
// then, when traversing duplications, the paralogy ring of that top segment (mapSelf top branch,
// halSegmentMapper.cpp:265-288; toNextParalogy, halTopSegmentIterator.cpp:99-107: follow paralogyIndex,
// flip strand iff the two segments' parentReversed differ; emit-then-test do/while).
// FINAL (the hop into the target genome): the pieces leave as final mapped pieces instead of (index, offset) pieces — the
// child segment's start and length come with the DownRec, so the forward target coordinate costs nothing, and the
// per-interval piece count is taken here (when a piece is stored); this replaces a k_finalize pass over the
// frontier (29 B written and read back per piece plus a gather).  In that form a frontier entry carries the forward
// source start in sPos and the forward target start in so (k_scatter_front turns it into a MappedRec).
template <typename C, bool FINAL>
__global__ void __launch_bounds__(256) k_down_ring(const DownRec<C> *__restrict__ down, const TopRec<C> *__restrict__ ctop, Frontier in,
                                                   const unsigned long long *inCount, uint32_t cap, Frontier out,
                                                   unsigned long long *outCount, int64_t minLength, int doDupes,
                                                   unsigned long long *counters, unsigned long long *kstat, uint32_t *__restrict__ perQuery) {
    __shared__ FrontView fview;
    const uint32_t n = front_view_init(&fview, inCount, cap);
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t topDerefs = 0, botDerefs = 0;
    __shared__ StageMem stageMem;
    Stage stage;
    stage.init(&stageMem, out, outCount, counters, cap);
    if (FINAL)
        stage.perQuery = perQuery;
    for (uint32_t base = wave * 64; base < n; base += wavesTotal * 64) {
        const uint32_t li = base + lane_id();
        const uint32_t i = li < n ? front_slot(&fview, li, cap) : 0;
        bool act = false;
        int32_t qid = 0, so = 0, len = 0, t0 = 0, cur = 0;
        int64_t sPos = 0;
        uint8_t fl = 0;
        int32_t rcPar = -1, rcEnc = 0; // paralogy link and parentEnc (only its strand bit is used) of `cur`
        int64_t rcStart = 0, rcLen = 0; // start and length of `cur` (ring members have one length: they share a parent segment)
        if (li < n) {
            const int32_t b = in.idx[i];
            len = in.len[i];
            const DownRec<C> d = down[b]; // link, start, length and paralogy of the child segment in one gather
            ++botDerefs;
            if (d.childEnc >= 0 && (int64_t)len >= minLength) {
                act = true;
                qid = in.qid[i];
                sPos = in.sPos[i];
                so = (int32_t)in.so[i];
                fl = in.flags[i];
                if (d.childEnc & 1)
                    fl ^= F_TREV;
                t0 = cur = d.childEnc >> 1;
                rcPar = d.paralogy;
                rcEnc = d.childEnc; // a top segment's parentReversed equals its parent's childReversed
                rcStart = (int64_t)d.childStart;
                rcLen = (int64_t)d.len;
                if (doDupes)
                    ++topDerefs; // the ring walk's look at the child segment
            }
        }
        while (__any(act)) {
            if (FINAL) {
                int64_t tLo = 0, sLo = 0;
                if (act) {
                    // SegmentIterator::getStartPosition / getEndPosition in forward coordinates (halSegmentIterator.cpp:46-67)
                    tLo = !(fl & F_TREV) ? rcStart + so : rcStart + rcLen - so - len;
                    sLo = !(fl & F_SREV) ? sPos : sPos - len + 1;
                    ++topDerefs; // k_finalize's share
                }
                stage.emit(act, qid, sLo, 0, tLo, len, fl);
            } else {
                stage.emit(act, qid, sPos, cur, so, len, fl);
            }
            if (act) {
                if (!doDupes) {
                    act = false;
                } else {
                    if (rcPar < 0) {
                        act = false; // no next paralogy: the do/while exits after the first emit
                    } else {
                        const TopRec<C> nr = ctop[rcPar];
                        ++topDerefs;
                        if ((nr.parentEnc & 1) != (rcEnc & 1))
                            fl ^= F_TREV;
                        cur = rcPar;
                        rcPar = nr.paralogy;
                        rcEnc = nr.parentEnc;
                        rcStart = (int64_t)nr.start;
                        act = rcPar >= 0 && cur != t0; // while (hasNextParalogy && index != start)
                    }
                }
            }
        }
    }
    stage.flush();
    stat_add(&kstat[0], topDerefs);
    stat_add(&kstat[1], botDerefs);
}

// ---------------------------------------------------------------------------------------------
// PD / PU: a piece on one tiling of genome G -> pieces on G's other tiling.
// PD, top -> bottom: mapDown, top branch (halSegmentMapper.cpp:144-184) with BottomSegmentIterator::toParseDown
// (api/impl/halBottomSegmentIterator.cpp:51-76).  PU, bottom -> top: mapUp / mapSelf, bottom branch
// (halSegmentMapper.cpp:40-78, :289-328) with TopSegmentIterator::toParseUp (api/impl/halTopSegmentIterator.cpp:55-81).
// Both: start at the parse index, scan right to the segment holding the piece's first base, then one output piece per
// overlapped segment (the toRight(rightCutoff) loop); the source side is sliced by the same deltas.
template <typename C> __device__ __forceinline__ int32_t parse_link(const TopRec<C> &r) {
    return r.botParse;
}
template <typename C> __device__ __forceinline__ int32_t parse_link(const BotRec<C> &r) {
    return r.topParse;
}
// FROM_TOP: statistics slot of the `from` table (0 = top records, 1 = bottom records)
template <typename FROM, typename TO, int FROM_SLOT>
__device__ __forceinline__ void parse_body(const FROM *__restrict__ from, const TO *__restrict__ to, Frontier in,
                                           const unsigned long long *inCount, uint32_t cap, Frontier out, unsigned long long *outCount,
                                           unsigned long long *counters, unsigned long long *kstat, StageMem *stageMem, FrontView *fview) {
    const uint32_t n = front_view_init(fview, inCount, cap);
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t fromDerefs = 0, toDerefs = 0;
    Stage stage;
    stage.init(stageMem, out, outCount, counters, cap);
    for (uint32_t base = wave * 64; base < n; base += wavesTotal * 64) {
        const uint32_t li = base + lane_id();
        const uint32_t i = li < n ? front_slot(fview, li, cap) : 0;
        bool act = li < n;
        int32_t qid = 0;
        int64_t sPos = 0, lo = 0, hi = -1;
        uint8_t fl = 0;
        int64_t j = 0, curStart = 0;
        if (act) {
            const int32_t t = in.idx[i];
            const int32_t so = (int32_t)in.so[i], len = in.len[i];
            qid = in.qid[i];
            sPos = in.sPos[i];
            fl = in.flags[i];
            const FROM tr = from[t];
            ++fromDerefs;
            if (!(fl & F_TREV))
                lo = (int64_t)tr.start + so;
            else
                lo = (int64_t)from[t + 1].start - so - len;
            hi = lo + len - 1;
            j = parse_link(tr);
            ++toDerefs;
            for (;;) {
                const int64_t nextStart = (int64_t)to[j + 1].start;
                if (nextStart > lo)
                    break;
                ++j;
                ++toDerefs;
            }
            curStart = (int64_t)to[j].start;
        }
        while (__any(act)) {
            bool emit = false;
            int32_t oSo = 0, oLen = 0;
            int64_t oSPos = 0;
            int64_t nextStart = 0;
            if (act) {
                nextStart = (int64_t)to[j + 1].start;
                const int64_t plo = lo > curStart ? lo : curStart;
                const int64_t phi = hi < nextStart - 1 ? hi : nextStart - 1;
                oLen = (int32_t)(phi - plo + 1);
                emit = true;
                int64_t d;
                if (!(fl & F_TREV)) {
                    oSo = (int32_t)(plo - curStart);
                    d = plo - lo;
                } else {
                    oSo = (int32_t)(nextStart - 1 - phi);
                    d = hi - phi;
                }
                oSPos = (fl & F_SREV) ? sPos - d : sPos + d;
            }
            stage.emit(emit, qid, oSPos, (int32_t)j, oSo, oLen, fl);
            if (act) {
                ++j;
                curStart = nextStart;
                act = curStart <= hi;
                if (act)
                    ++toDerefs;
            }
        }
    }
    stage.flush();
    stat_add(&kstat[FROM_SLOT], fromDerefs);
    stat_add(&kstat[1 - FROM_SLOT], toDerefs);
}

template <typename C>
__global__ void __launch_bounds__(256) k_parse_down(const TopRec<C> *__restrict__ top, const BotRec<C> *__restrict__ bot, Frontier in,
                                                    const unsigned long long *inCount, uint32_t cap, Frontier out,
                                                    unsigned long long *outCount, unsigned long long *counters,
                                                    unsigned long long *kstat) {
    __shared__ FrontView fview;
    __shared__ StageMem stageMem;
    parse_body<TopRec<C>, BotRec<C>, 0>(top, bot, in, inCount, cap, out, outCount, counters, kstat, &stageMem, &fview);
}

template <typename C>
__global__ void __launch_bounds__(256) k_parse_up(const BotRec<C> *__restrict__ bot, const TopRec<C> *__restrict__ top, Frontier in,
                                                  const unsigned long long *inCount, uint32_t cap, Frontier out,
                                                  unsigned long long *outCount, unsigned long long *counters,
                                                  unsigned long long *kstat) {
    __shared__ FrontView fview;
    __shared__ StageMem stageMem;
    parse_body<BotRec<C>, TopRec<C>, 1>(bot, top, in, inCount, cap, out, outCount, counters, kstat, &stageMem, &fview);
}

// ---------------------------------------------------------------------------------------------
// R: top piece -> itself and the other members of its paralogy ring (mapSelf, top branch, halSegmentMapper.cpp:265-288;
// toNextParalogy, halTopSegmentIterator.cpp:99-107).  The do/while emits before it tests, follows the ring while the
// segment it moved to has a next paralogy, the piece is at least minLength long and the walk is not back at its start.
// (k_down_ring carries the same loop after its child hop; this kernel is the stand-alone form mapRecursiveParalogies needs.)
template <typename C>
__global__ void __launch_bounds__(256) k_ring(const TopRec<C> *__restrict__ top, Frontier in, const unsigned long long *inCount, uint32_t cap,
                                              Frontier out, unsigned long long *outCount, int64_t minLength, unsigned long long *counters,
                                              unsigned long long *kstat) {
    __shared__ FrontView fview;
    const uint32_t n = front_view_init(&fview, inCount, cap);
    const uint32_t wavesTotal = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t topDerefs = 0;
    __shared__ StageMem stageMem;
    Stage stage;
    stage.init(&stageMem, out, outCount, counters, cap);
    for (uint32_t base = wave * 64; base < n; base += wavesTotal * 64) {
        const uint32_t li = base + lane_id();
        const uint32_t i = li < n ? front_slot(&fview, li, cap) : 0;
        bool act = li < n;
        int32_t qid = 0, len = 0, t0 = 0, cur = 0;
        int64_t sPos = 0, so = 0;
        uint8_t fl = 0;
        int32_t rcPar = -1, rcEnc = 0;
        if (act) {
            qid = in.qid[i];
            sPos = in.sPos[i];
            so = in.so[i];
            len = in.len[i];
            fl = in.flags[i];
            t0 = cur = in.idx[i];
            const TopRec<C> rc = top[cur];
            ++topDerefs;
            rcPar = rc.paralogy;
            rcEnc = rc.parentEnc;
        }
        while (__any(act)) {
            stage.emit(act, qid, sPos, cur, so, len, fl);
            if (act) {
                if (rcPar < 0) {
                    act = false;
                } else {
                    const TopRec<C> nr = top[rcPar];
                    ++topDerefs;
                    if ((nr.parentEnc & 1) != (rcEnc & 1))
                        fl ^= F_TREV;
                    cur = rcPar;
                    rcPar = nr.paralogy;
                    rcEnc = nr.parentEnc;
                    act = rcPar >= 0 && (int64_t)len >= minLength && cur != t0;
                }
            }
        }
    }
    stage.flush();
    stat_add(&kstat[0], topDerefs);
}

// ---------------------------------------------------------------------------------------------
// Final: pieces in the target genome -> forward coordinates, and count pieces per query.
// Positions per SegmentIterator::getStartPosition/getEndPosition (halSegmentIterator.cpp:46-67).
// runs of equal interval index among the lanes of a wavefront (neighbouring pieces usually belong to one interval, the
// more so once the batch is sorted): one atomic per run instead of one per lane.  q < 0 marks an idle lane.
// Returns the run's first lane and its length.
__device__ __forceinline__ void wave_run_of(int32_t q, int &start, int &len) {
    const int lane = lane_id();
    const int32_t prev = __shfl_up(q, 1);
    const bool leader = lane == 0 || q != prev;
    const unsigned long long mask = __ballot(leader);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    start = 63 - __clzll((long long)(mask & upto));
    const unsigned long long above = mask & ~upto;
    const int end = above ? __ffsll((long long)above) - 1 : 64;
    len = end - start;
}

template <typename REC>
__global__ void __launch_bounds__(256) k_finalize(const REC *__restrict__ segs, Frontier in, const unsigned long long *inCount, uint32_t cap,
                                                  Mapped out, uint32_t *__restrict__ perQuery, unsigned long long *counters,
                                                  unsigned long long *kstat, int isTop) {
    __shared__ FrontView fview;
    const uint32_t n = front_view_init(&fview, inCount, cap);
    uint32_t derefs = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) { // whole wavefronts iterate together
        const uint32_t i = base + threadIdx.x;
        int32_t q = -1;
        if (i < n) {
            const uint32_t p = front_slot(&fview, i, cap);
            const int32_t idx = in.idx[p];
            const int32_t so = (int32_t)in.so[p], len = in.len[p];
            const uint8_t fl = in.flags[p];
            const int64_t sPos = in.sPos[p];
            q = in.qid[p];
            int64_t lo;
            if (!(fl & F_TREV))
                lo = (int64_t)segs[idx].start + so;
            else
                lo = (int64_t)segs[idx + 1].start - so - len;
            ++derefs;
            MappedRec r;
            r.tLo = lo;
            r.sLo = !(fl & F_SREV) ? sPos : sPos - len + 1;
            r.len = len;
            r.qid = q;
            r.flags = fl;
            r._pad = 0;
            out.rec[i] = r;
        }
        int start, len;
        wave_run_of(q, start, len);
        if (q >= 0 && lane_id() == start)
            atomicAdd(&perQuery[q], (uint32_t)len);
    }
    stat_add(&kstat[isTop ? 0 : 1], derefs);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        counters[CNT_MAPPED] = n;
}

// group pieces by query: slot = offset[q] + cursor[q]++
static __global__ void __launch_bounds__(256) k_scatter(Mapped in, const unsigned long long *inCount, uint32_t cap, const uint32_t *__restrict__ offset,
                                                 uint32_t *__restrict__ cursor, Mapped out) {
    const uint32_t n = (uint32_t)min((unsigned long long)cap, *inCount); // CNT_MAPPED: dense count written by k_finalize
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        MappedRec r;
        r.qid = -1;
        if (i < n)
            r = in.rec[i];
        int start, len;
        wave_run_of(r.qid, start, len);
        uint32_t b = 0;
        if (r.qid >= 0 && lane_id() == start)
            b = atomicAdd(&cursor[r.qid], (uint32_t)len);
        b = __shfl(b, start);
        if (i < n)
            out.rec[offset[r.qid] + b + (uint32_t)(lane_id() - start)] = r;
    }
}

// the same from a frontier of FINAL pieces (k_down_ring<C, true>)
static __global__ void __launch_bounds__(256) k_scatter_front(Frontier in, const unsigned long long *inCount, uint32_t cap,
                                                              const uint32_t *__restrict__ offset, uint32_t *__restrict__ cursor, Mapped out,
                                                              unsigned long long *counters) {
    __shared__ FrontView fview;
    const uint32_t n = front_view_init(&fview, inCount, cap);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        MappedRec r;
        r.qid = -1;
        if (i < n) {
            const uint32_t p = front_slot(&fview, i, cap);
            r.tLo = in.so[p];
            r.sLo = in.sPos[p];
            r.len = in.len[p];
            r.qid = in.qid[p];
            r.flags = in.flags[p];
            r._pad = 0;
        }
        int start, len;
        wave_run_of(r.qid, start, len);
        uint32_t b = 0;
        if (r.qid >= 0 && lane_id() == start)
            b = atomicAdd(&cursor[r.qid], (uint32_t)len);
        b = __shfl(b, start);
        if (i < n)
            out.rec[offset[r.qid] + b + (uint32_t)(lane_id() - start)] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        counters[CNT_MAPPED] = n;
}

} // namespace hgx
