// The single-pass liftover kernels over the MERGED table (hgx_merged_kernels.hpp): BlockLiftover::liftInterval
// (liftover/impl/halBlockLiftover.cpp:46-113) for a batch of intervals — toSite, the per-source-segment halMapSegment calls,
// insertAndBreakOverlaps, extractSegment, the stable sort on the source start (liftover/impl/halLiftover.cpp:90) — as
//   k_lift_classify   per interval: one look at the bucket table (where its records start) and a walk over its records that
//                     counts its output lines and finds out whether it must go the general way — hgx_finish_kernel.hpp, over
//                     the unmerged table, which the wavefront that meets such an interval does on the spot; per wavefront:
//                     the lines of its 64 intervals;
//   k_lift_totals     one workgroup: the lines of every group of 64 tiles, all lines, the statistics, the host's report;
//   k_lift_merged     reads the records a second time, clips them, orders them, and writes the hgx_records of the whole batch
//                     densely and in input order — once.
// An unflagged interval's records have pairwise disjoint target ranges, so (see hgx_merged_kernels.hpp) its output lines
// are exactly its records clipped to it.  The reference prints them stably sorted by source start, ties in target order:
// the table is sorted by (source start, target start), so the clipped records already come in that order except for the
// ones that begin at or before the interval's first base — they all start at that base after clipping and are ordered by
// target start among themselves (they are the first of the interval's records; usually there is one).
//
// Dense output in input order needs every interval's offset = the number of lines of all intervals before it.  The count
// and the store are two launches, so the store knows every count when it starts: a workgroup adds up the counts of the groups
// before its tile's group and of the wavefronts before its tile in the group (two reads, DPP sums) — no waiting.  (The
// one-launch form of this — count, publish, decoupled look-back, store — spent 44 % of its wavefronts' cycles waiting for
// the slowest of the 64 tiles in front: profiles/r02k_notes.md.)
#pragma once
#include "../../include/hgx.h"
#include "hgx_finish_kernel.hpp"

namespace hgx {

static constexpr int LIFT_TILE = 256;        // intervals per tile (= threads per workgroup; LIFT_TILE_SHIFT in hgx_finish_kernel.hpp)
static constexpr uint32_t LIFT_MAX_BOUND = 64; // records in reach of an interval the wave-wide rounds can hold
static constexpr uint32_t KB_GENERAL = 0x80000000u;
enum { CNT_LIFT_TOTAL = 1 }; // counters[] slot (hgx_liftover_kernels.hpp uses 0 and 3..7)
static_assert(STAT_LINES <= 64, "k_lift_totals folds the statistics copies with one lane each");
static_assert(LIFT_TILE == 1 << LIFT_TILE_SHIFT, "the finishing kernels add late counts to the tile totals");

// ---- wave-wide scans on the VALU: DPP row shifts and row broadcasts (gfx9 encodings), no trip through the LDS crossbar ----
// (a ds_bpermute-based scan is six dependent LDS-pipe round trips; the kernel runs several per tile)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t lift_dpp(uint32_t ident, uint32_t v) {
    // lanes whose source lies outside their row (or whose row is masked off) keep `ident`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROW_MASK, 0xF, false);
}
struct LiftSum {
    static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a + b; }
};
struct LiftMax {
    static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a > b ? a : b; }
};
// inclusive scan over the 64 lanes (identity 0 for both operations used here)
template <typename Op> __device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v = Op::f(v, lift_dpp<0x111, 0xF>(0u, v)); // row_shr:1
    v = Op::f(v, lift_dpp<0x112, 0xF>(0u, v)); // row_shr:2
    v = Op::f(v, lift_dpp<0x114, 0xF>(0u, v)); // row_shr:4
    v = Op::f(v, lift_dpp<0x118, 0xF>(0u, v)); // row_shr:8    -> every row of 16 scanned
    v = Op::f(v, lift_dpp<0x142, 0xA>(0u, v)); // row_bcast:15 -> rows 1 and 3 add the row before them
    v = Op::f(v, lift_dpp<0x143, 0xC>(0u, v)); // row_bcast:31 -> rows 2 and 3 add the lower half
    return v;
}
__device__ __forceinline__ uint32_t wave_total(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan<LiftSum>(v), 63);
}
// sum of 64 values below 2^62, in three limbs of at most 24 bits (64 of them stay below 2^30)
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
    const unsigned long long a = wave_total((uint32_t)v & 0xFFFFFFu), b = wave_total((uint32_t)(v >> 24) & 0xFFFFFFu),
                             c = wave_total((uint32_t)(v >> 48));
    return a + (b << 24) + (c << 48);
}

// ---------------------------------------------------------------------------------------------
// quad helpers (DPP quad_perm: no LDS crossbar)
template <int CTRL> __device__ __forceinline__ uint32_t quad_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
static constexpr int QUAD_XOR1 = 0xB1, QUAD_XOR2 = 0x4E, QUAD_LANE3 = 0xFF; // quad_perm:[1,0,3,2], [2,3,0,1], [3,3,3,3]
struct LiftMin {
    static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a < b ? a : b; }
};
template <typename Op> __device__ __forceinline__ uint32_t quad_reduce(uint32_t v) {
    v = Op::f(v, quad_dpp<QUAD_XOR1>(v));
    return Op::f(v, quad_dpp<QUAD_XOR2>(v));
}
struct LiftOr {
    static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a | b; }
};

// what one lane of a quad has seen of its interval's records
template <typename C> struct LiftScan {
    uint32_t cnt = 0, flags = 0, first = 0xFFFFFFFFu, last = 0;
    bool cont = false; // the quad's fourth record still begins inside the interval: four more
    __device__ __forceinline__ void take(const ComposedRec<C> &r, uint32_t idx, C gs, C ge) {
        const bool more = r.sLo <= ge;
        const bool ov = more && r.sLo + r.len - 1 >= gs;
        if (ov) {
            ++cnt;
            flags |= (r.mEncF >> 1) & 1u;
            first = idx < first ? idx : first;
            last = idx > last ? idx : last;
        }
        cont = quad_dpp<QUAD_LANE3>(more ? 1u : 0u) != 0;
    }
};
// what the quads need to know of an interval: where its records start, its first and last base (one ds_read_b128 with 32-bit
// coordinates, two with 64-bit ones)
template <typename C> struct alignas(16) LiftAsk {
    C gs, ge;
    uint32_t k0, _pad;
};

// k_lift_general_list, in front of k_lift_classify when a plan's last run had few general intervals: which intervals must go the
// general way, found from the batch alone — bits: a bit per bucket of the merged table, set where a flagged record touches the
// bucket (hgx_merged_kernels.hpp: k_bucket_flag_bits); an interval with a flagged record among its own has such a bucket among
// the ones it touches.  Yes as well for intervals longer than the table's window (general whatever their records), and for the
// few that touch more buckets than two words of bits cover.  mask[q / 64]: those intervals as bits; list: the same as LIFT_LISTS
// lists of listCap entries each, counts[l * LIFT_LIST_PITCH] long (a workgroup appends to the list of its index: one counter for
// all of them is a queue of same-address atomics at the memory side, 15 ns each).  k_lift_classify's workers take the lists, its
// tiles leave the intervals of the mask alone.  waveExtra / groupExtra: see k_lift_classify.
static constexpr uint32_t LIFT_LISTS = 64, LIFT_LIST_PITCH = 16; // (counters 128 bytes apart)
// the rule of the list for one interval, from the interval and the bucket bits alone (k_lift_general_list's, two dependent loads):
// the scouts of k_lift_classify find their intervals with it and the tiles tell by it which intervals are not theirs — the same
// function of the same words on both sides, so nothing has to pass between them
__device__ __forceinline__ bool lift_listed(int64_t gs, int64_t ge, int64_t genomeLength, const uint32_t *__restrict__ bits, int shift,
                                            int64_t window) {
    const bool valid = ge >= gs && gs >= 0 && gs < genomeLength;
    if (!valid)
        return false;
    const int64_t b0 = gs >> shift, nbk = ((ge < genomeLength ? ge : genomeLength - 1) >> shift) - b0 + 1;
    if (ge - gs >= window || nbk > 32)
        return true;
    const uint64_t both = ((uint64_t)bits[(b0 >> 5) + 1] << 32) | bits[b0 >> 5];
    return (((uint32_t)(both >> (b0 & 31))) & (nbk >= 32 ? 0xFFFFFFFFu : ((1u << nbk) - 1u))) != 0;
}
static constexpr uint32_t LIFT_SCOUT_ROUND = 1024; // intervals a scout workgroup looks at before it finishes what it found among them
static __global__ void __launch_bounds__(256) k_lift_general_list(const int64_t *__restrict__ gStart, const int64_t *__restrict__ gEnd, uint32_t nq,
                                                                  int64_t genomeLength, const uint32_t *__restrict__ bits, int shift, int64_t window,
                                                                  unsigned long long *__restrict__ mask, uint32_t *__restrict__ list, uint32_t listCap,
                                                                  unsigned long long *__restrict__ counts, uint32_t *__restrict__ waveExtra,
                                                                  unsigned long long *__restrict__ groupExtra) {
    // two intervals per thread (256 apart), their loads issued together: the kernel is two dependent round trips to memory and
    // nothing else, and this way all its wavefronts are resident at once
    const int lane = lane_id();
    uint32_t q[2];
    int64_t gs[2], ge[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        q[u] = blockIdx.x * 512u + (uint32_t)u * 256u + threadIdx.x;
        gs[u] = 0;
        ge[u] = -1;
        if (q[u] < nq) {
            gs[u] = gStart[q[u]];
            ge[u] = gEnd[q[u]];
        }
    }
    bool mine[2];
    uint32_t w0[2], w1[2];
    int64_t b0[2], nbk[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const bool valid = ge[u] >= gs[u] && gs[u] >= 0 && gs[u] < genomeLength;
        b0[u] = valid ? gs[u] >> shift : 0;
        nbk[u] = valid ? ((ge[u] < genomeLength ? ge[u] : genomeLength - 1) >> shift) - b0[u] + 1 : 0;
        mine[u] = valid && (ge[u] - gs[u] >= window || nbk[u] > 32);
        w0[u] = w1[u] = 0;
        if (valid && !mine[u]) {
            w0[u] = bits[b0[u] >> 5];
            w1[u] = bits[(b0[u] >> 5) + 1];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (nbk[u] > 0 && !mine[u]) {
            const uint64_t both = ((uint64_t)w1[u] << 32) | w0[u];
            mine[u] = (((uint32_t)(both >> (b0[u] & 31))) & (nbk[u] >= 32 ? 0xFFFFFFFFu : ((1u << nbk[u]) - 1u))) != 0;
        }
        // (the words the workers count their lines in, per 64 intervals and per group of 64 tiles: cleared here, in front of them)
        if (q[u] < nq && lane == 0)
            waveExtra[q[u] >> 6] = 0;
        if (q[u] < nq && (q[u] & ((64u << LIFT_TILE_SHIFT) - 1u)) == 0)
            groupExtra[q[u] >> (6 + LIFT_TILE_SHIFT)] = 0;
        const unsigned long long m = __ballot(mine[u]);
        if (q[u] < nq && lane == 0)
            mask[q[u] >> 6] = m;
        if (m) {
            const uint32_t l = blockIdx.x % LIFT_LISTS;
            unsigned long long at = 0;
            if (lane == 0)
                at = atomicAdd(&counts[l * LIFT_LIST_PITCH], (unsigned long long)__popcll(m));
            at = __shfl(at, 0);
            if (mine[u]) // (a list cannot run full: it holds every interval its workgroups look at)
                list[(size_t)l * listCap + at + __popcll(m & ((1ull << lane) - 1ull))] = q[u];
        }
    }
}

// k_lift_classify, a workgroup per tile: kb[q] = {first record that overlaps interval q, number of records from there to the
// last one that overlaps | KB_GENERAL}, nOut[q] = the interval's lines, waveTotal[q / 64] = the lines of a wavefront's 64
// intervals.
// An interval's records are found by one look at the bucket table (the first record that touches the bucket of its first
// base) and a scan from there to the first record that begins behind its last base (the table ends in sentinels).  Four
// lanes scan for one interval, four records — one 64-byte line — per trip: the kernel is bound by the number of separate
// lines its gathers touch, and this way an interval costs one or two besides the bucket entry.
// General intervals (an overlapping record that carries the flag, more than LIFT_MAX_BOUND records, longer than the table's
// window):
//   INLINE: finished on the spot by the wavefront that meets them (general_interval, hgx_finish_kernel.hpp: the reference's
//   general algorithm in registers over the unmerged table) — nOut[q] / offset[q] then say where k_lift_merged finds their
//   records; the ones that passes on (more than 64 pieces) are listed in lateList for k_locate_through + k_finish_lds, which
//   add their lines to the totals.  A general interval costs its own wavefront a handful of dependent memory round trips;
//   in a launch of their own the same round trips were the batch's critical path.
//   !INLINE (HGX_FINISH_WAVE=0, a cross-check): all of them are listed.
//   workers > 0 (INLINE): the first `workers` workgroups of the grid are not tiles; they finish the general intervals
//   k_lift_general_list found in the batch at the start of the launch (see there); the tiles leave those intervals alone (kb /
//   nOut / offset are the workers' to write; their lines are counted in waveExtra and groupExtra, beside the tiles' waveTotal:
//   k_lift_general_list has cleared them, k_lift_totals and k_lift_merged add them in).
template <typename C, bool INLINE, int MINW>
static __global__ void __launch_bounds__(256, MINW) k_lift_classify(const int64_t *__restrict__ gStart, const int64_t *__restrict__ gEnd,
                                                              const uint8_t *__restrict__ strand, uint32_t nq, int64_t genomeLength,
                                                              const uint32_t *__restrict__ coarse, int shift, int64_t window,
                                                              const ComposedRec<C> *__restrict__ recs, uint2 *__restrict__ kb,
                                                              GeneralTable<C> GT, unsigned long long *kstat, unsigned long long *kstatStore,
                                                              uint32_t *__restrict__ offset,
                                                              uint32_t *__restrict__ nOut, uint32_t *__restrict__ lateList,
                                                              unsigned long long *__restrict__ lateCount, uint32_t *__restrict__ waveTotal,
                                                              uint32_t workers, uint32_t *__restrict__ waveExtra,
                                                              const unsigned long long *__restrict__ workMask, const uint32_t *__restrict__ workList,
                                                              uint32_t workListCap, const unsigned long long *__restrict__ workCounts,
                                                              unsigned long long *__restrict__ groupExtra, uint32_t scoutShare,
                                                              const uint32_t *__restrict__ flagBits, uint32_t mLast, uint32_t *__restrict__ otherWaveExtra,
                                                              unsigned long long *__restrict__ otherGroupExtra) {
    __shared__ C sDAll[INLINE ? 4 : 1][INLINE ? 128 : 1];
    __shared__ uint8_t sOwnAll[INLINE ? 4 : 1][INLINE ? 64 : 1];
    __shared__ LiftAsk<C> sAsk[4][64];
    __shared__ uint4 sAnswer[4][64];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const int quad = lane >> 2, c = lane & 3;
    uint32_t generalSeen = 0, used = 0, generalLines = 0;
    const uint32_t nTiles = (nq + (uint32_t)LIFT_TILE - 1) >> LIFT_TILE_SHIFT;
    LIFT_PROF_DECL;
    if (INLINE && blockIdx.x < workers && scoutShare != 0) {
        // ---- a scout (round 6): a worker that finds its general intervals itself.  The workgroups in front of the grid each look at
        // a share of the batch — the interval's two ends and two words of the bucket bits, lift_listed: four intervals a thread, their
        // loads issued together —, put what they find in a list in LDS and finish it, a wavefront an interval, round-robin.  No
        // launch in front (k_lift_general_list's pass over the batch cost what the tail it removed did, and more once batches
        // overlap: DESIGN 4.0.2, 4.0.3), nothing passed between workgroups: a scout takes an interval exactly when the tile's own
        // scan calls it general for a flag or its length (the same test on both sides).  Every general interval of the batch is
        // under way three microseconds into the launch.
        uint32_t *sList = (uint32_t *)&sAnswer[0][0]; // (1024 words: LIFT_SCOUT_ROUND)
        __shared__ uint32_t sListN;
        const uint32_t lo = blockIdx.x * scoutShare, hi = lo + scoutShare < nq ? lo + scoutShare : nq;
        for (uint32_t base = lo; base < hi; base += LIFT_SCOUT_ROUND) { // (usually one round: the host sizes the shares for it)
            if (threadIdx.x == 0)
                sListN = 0;
            __syncthreads();
            uint32_t sq[4];
            int64_t sgs[4], sge[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sq[u] = base + (uint32_t)u * 256u + threadIdx.x;
                sgs[u] = 0;
                sge[u] = -1;
                if (sq[u] < hi) {
                    sgs[u] = gStart[sq[u]];
                    sge[u] = gEnd[sq[u]];
                }
            }
            bool mine[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                mine[u] = lift_listed(sgs[u], sge[u], genomeLength, flagBits, shift, window);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long m = __ballot(mine[u]);
                if (m) {
                    uint32_t at = 0;
                    if (lane == 0)
                        at = atomicAdd(&sListN, (uint32_t)__popcll(m));
                    at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                    if (mine[u])
                        sList[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = sq[u];
                }
            }
            __syncthreads();
            const uint32_t nWork = sListN;
            for (uint32_t i = (uint32_t)w; i < nWork; i += 4u) {
                const uint32_t oq = (uint32_t)__builtin_amdgcn_readfirstlane((int)sList[i]);
                const int64_t os = gStart[oq], oe = gEnd[oq];
                {   // the tile's own question, put to the interval's first 64 merged records at once: is it general for a flag or for its
                    // length, with a record?  Only then is it the scout's (the bits say "may be": a quarter of the listed are not)
                    const C gsC = (C)os, geC = (C)(oe < genomeLength ? oe : genomeLength - 1);
                    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)coarse[os >> shift]);
                    const uint32_t idx = k0 + (uint32_t)lane < mLast ? k0 + (uint32_t)lane : mLast;
                    const ComposedRec<C> r = recs[idx];
                    const bool ov = r.sLo <= geC && r.sLo + r.len - 1 >= gsC;
                    const bool take = __ballot(ov) != 0 && (oe - os >= window || __ballot(ov && ((r.mEncF >> 1) & 1u) != 0) != 0);
                    if (!take)
                        continue;
                }
                int nl = 0;
                uint32_t gbase = 0;
                const int rc = general_interval<C>(lane, GT, oq, os, oe, strand[oq], sDAll[w], sOwnAll[w], used, nl, gbase LIFT_PROF_ARG);
                if (lane == 0) {
                    if (rc == 0)
                        lateList[atomicAdd(lateCount, 1ull)] = oq;
                    kb[oq] = make_uint2(0u, KB_GENERAL);
                    offset[oq] = gbase;
                    nOut[oq] = (uint32_t)nl;
                    if (nl > 0) {
                        atomicAdd(&waveExtra[oq >> 6], (uint32_t)nl);
                        atomicAdd(&groupExtra[oq >> (6 + LIFT_TILE_SHIFT)], (unsigned long long)nl);
                    }
                }
                generalSeen += lane == 0 ? 1u : 0u;
                generalLines += lane == 0 ? (uint32_t)nl : 0u;
            }
            __syncthreads();
        }
        stat_add(&kstat[0], used);
        stat_add(&kstat[1], generalSeen);
        stat_add(&kstatStore[1], generalLines);
        stat_add(&GT.counters[CNT_DSTAT0 + STAT_MAPPED], used);
        return;
    }
    if (INLINE && blockIdx.x < workers) {
        // ---- a worker: the general intervals k_lift_general_list found, a wavefront each, dealt round-robin.  The workgroups in
        // front of the grid do this, so the general intervals of the whole batch are under way when the launch begins and not when
        // their tile comes up (the last tiles' would otherwise end the launch one general interval's latency — two, three dependent
        // round trips and finish_wave — after everything else).
        // (lane l holds the length of list l and what is in the lists before it)
        const uint32_t mine = (uint32_t)workCounts[(uint32_t)lane * LIFT_LIST_PITCH];
        const uint32_t incl = wave_incl_scan<LiftSum>(mine);
        const uint32_t nWork = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t i = blockIdx.x * 4u + (uint32_t)w; i < nWork; i += workers * 4u) {
            const int l = (int)__popcll(__ballot(incl <= i)); // the list item i is in
            const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), l);
            const uint32_t oq = (uint32_t)__builtin_amdgcn_readfirstlane((int)workList[(size_t)l * workListCap + (i - before)]);
            const int64_t os = gStart[oq], oe = gEnd[oq];
            int nl = 0;
            uint32_t base = 0;
            const int rc = general_interval<C>(lane, GT, oq, os, oe, strand[oq], sDAll[w], sOwnAll[w], used, nl, base LIFT_PROF_ARG);
            if (lane == 0) {
                if (rc == 0) // (its lines come later, or not at all: a run that does not make the launches behind this one is repeated)
                    lateList[atomicAdd(lateCount, 1ull)] = oq;
                kb[oq] = make_uint2(0u, KB_GENERAL);
                offset[oq] = base;
                nOut[oq] = (uint32_t)nl;
                if (nl > 0) { // (the interval's tile counts its 64 without it)
                    atomicAdd(&waveExtra[oq >> 6], (uint32_t)nl);
                    atomicAdd(&groupExtra[oq >> (6 + LIFT_TILE_SHIFT)], (unsigned long long)nl);
                }
            }
            generalSeen += lane == 0 ? 1u : 0u;
            generalLines += lane == 0 ? (uint32_t)nl : 0u;
        }
        stat_add(&kstat[0], used);
        stat_add(&kstat[1], generalSeen);
        stat_add(&kstatStore[1], generalLines);
        stat_add(&GT.counters[CNT_DSTAT0 + STAT_MAPPED], used);
        return;
    }
    const uint32_t firstTile = INLINE ? blockIdx.x - workers : blockIdx.x;
    for (uint32_t tile = firstTile; tile < nTiles; tile += gridDim.x - (INLINE ? workers : 0u)) {
        const uint32_t q = tile * (uint32_t)LIFT_TILE + threadIdx.x;
        int64_t gs = 0, ge = -1;
        if (q < nq) {
            gs = gStart[q];
            ge = gEnd[q];
        }
        const bool valid = ge >= gs && gs >= 0 && gs < genomeLength;
        // (an interval that is not valid asks for nothing: no record begins at or before base -1)
        const uint32_t k0 = valid ? coarse[gs >> shift] : 0u;
        // (the scan stops at the first record that begins behind geC: below the sentinels' start whatever the interval says)
        const C gsC = valid ? (C)gs : (C)0, geC = valid ? (C)(ge < genomeLength ? ge : genomeLength - 1) : (C)-1;
        sAsk[w][lane] = LiftAsk<C>{gsC, geC, k0, 0u};
        wave_lds_fence();
        LIFT_PROF(0) // the intervals and their bucket entries have arrived
        // ---- the scan: quad `quad` of round `it` works for the interval of lane 16 * it + quad ----
        LiftAsk<C> ask[4];
        ComposedRec<C> r[4];
        LiftScan<C> sc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            ask[it] = sAsk[w][16 * it + quad];
            r[it] = recs[ask[it].k0 + (uint32_t)c];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            sc[it].take(r[it], ask[it].k0 + (uint32_t)c, ask[it].gs, ask[it].ge);
        for (uint32_t trip = 1; trip < LIFT_MAX_BOUND / 4 && __any(sc[0].cont || sc[1].cont || sc[2].cont || sc[3].cont); ++trip) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (sc[it].cont)
                    r[it] = recs[ask[it].k0 + 4u * trip + (uint32_t)c];
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (sc[it].cont) // (quad-uniform)
                    sc[it].take(r[it], ask[it].k0 + 4u * trip + (uint32_t)c, ask[it].gs, ask[it].ge);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t n = quad_reduce<LiftSum>(sc[it].cnt), fl = quad_reduce<LiftOr>(sc[it].flags | (sc[it].cont ? 2u : 0u)),
                           first = quad_reduce<LiftMin>(sc[it].first), last = quad_reduce<LiftMax>(sc[it].last);
            if (c == 0)
                sAnswer[w][16 * it + quad] = make_uint4(n, fl, first, last);
        }
        wave_lds_fence();
        const uint4 ans = sAnswer[w][lane];
        uint32_t cnt = ans.x;
        // flags: 1 a record that overlaps the interval carries the table's flag, 2 the scan stopped at LIFT_MAX_BOUND records
        // that begin inside the interval.  (Putting the flag's question — do two of the interval's records overlap on the
        // target? — to the records of a quad themselves spares a quarter of the general intervals at cfg2 and costs every
        // interval more than that saves: profiles/r02k_notes.md.)
        const bool general = valid && (cnt > 0 || (ans.y & 2u)) && ((ans.y & 3u) != 0 || ge - gs >= window);
        // (with workers: the intervals they take — every one with a flagged record among its own, and then some — are theirs
        // to answer; what is left for the tile is an interval with more records than the scan holds, which is listed.  Scouts take
        // exactly the general intervals with a record and a flag or the window's length: what the tile has just found out itself —
        // a test by the bucket bits here, two more gathers an interval, cost the launch ten microseconds)
        const bool theirs = INLINE && workers != 0 && q < nq &&
                            (scoutShare != 0 ? general && cnt > 0 && ((ans.y & 1u) != 0 || ge - gs >= window) : ((workMask[q >> 6] >> lane) & 1ull) != 0);
        if (q < nq && !theirs)
            kb[q] = general ? make_uint2(0u, KB_GENERAL) : cnt ? make_uint2(ans.z, ans.w - ans.z + 1u) : make_uint2(0u, 0u);
        generalSeen += general && !theirs ? 1u : 0u;
        LIFT_PROF(1) // counted
        if (INLINE && workers != 0) {
            if (general && !theirs)
                lateList[atomicAdd(lateCount, 1ull)] = q;
            if (general || theirs)
                cnt = 0;
        } else if (INLINE) {
            unsigned long long gm = __ballot(general);
            while (gm) { // (rare: about one wavefront in fifteen meets one at cfg2)
                const int o = __ffsll((long long)gm) - 1;
                gm &= gm - 1;
                const uint32_t oq = (uint32_t)wave_read<int32_t>((int32_t)q, o); // (readlane: wave-uniform values keep general_interval's loops scalar)
                const int64_t os = wave_read<int64_t>(gs, o), oe = wave_read<int64_t>(ge, o);
                int nl = 0;
                uint32_t base = 0;
                const int rc = general_interval<C>(lane, GT, oq, os, oe, strand[oq], sDAll[w], sOwnAll[w], used, nl, base LIFT_PROF_ARG);
                if (lane == 0) {
                    if (rc == 0) // (its lines come later, or not at all: a run that does not make the launches behind this one is repeated)
                        lateList[atomicAdd(lateCount, 1ull)] = oq;
                    offset[oq] = base;
                }
                if (lane == o) {
                    cnt = (uint32_t)nl;
                    generalLines += (uint32_t)nl;
                }
            }
        } else if (general) {
            lateList[atomicAdd(lateCount, 1ull)] = q;
            cnt = 0;
        }
        if (q < nq && !theirs)
            nOut[q] = cnt;
        // (no barrier: a wavefront that met a general interval does not keep the other three waiting)
        const uint32_t waveLines = wave_total(cnt);
        if (lane == 0)
            waveTotal[tile * 4u + (uint32_t)w] = waveLines;
        // (scouts: the words the next batch's scouts add to — the other set of two — are cleared by this batch's tiles)
        if (INLINE && otherWaveExtra) {
            if (lane == 0)
                otherWaveExtra[tile * 4u + (uint32_t)w] = 0;
            if ((tile & 63u) == 0 && threadIdx.x == 0)
                otherGroupExtra[tile >> 6] = 0;
        }
        LIFT_PROF(2) // general intervals done
        LIFT_PROF(3)
    }
    LIFT_PROF_FLUSH
    stat_add(&kstat[0], used);         // the "top" slot of this launch: unmerged records the general intervals clipped
    stat_add(&kstat[1], generalSeen);  // its "bottom" slot: general intervals
    // (the lines that come from merged records — nearly all — are not counted here: one or two atomics per wavefront on 32
    // cache lines cost the launch 1-3 us; k_lift_totals has them as all lines minus these)
    stat_add(&kstatStore[1], generalLines); // the "bottom" slot of the storing launch: lines that do not come from merged records
    stat_add(&GT.counters[CNT_DSTAT0 + STAT_MAPPED], used);
}

// what a record lane needs to know about the interval that owns its slot (one ds_read_b128 with 32-bit coordinates, two with
// 64-bit ones)
//   x: first record of the interval - its first slot (so that record = x + slot), y: its first slot,
//   32-bit: z: first base | minus strand << 31, w: last base | '.' strand << 31;  64-bit: the bases and the two bits apart
template <typename C> struct LiftIv;
template <> struct alignas(16) LiftIv<int32_t> {
    uint32_t x, y, z, w;
    static __device__ __forceinline__ LiftIv make(uint32_t x, uint32_t y, int32_t gs, int32_t ge, uint32_t minus, uint32_t dot) {
        return LiftIv{x, y, ((uint32_t)gs & 0x7FFFFFFFu) | (minus << 31), ((uint32_t)ge & 0x7FFFFFFFu) | (dot << 31)};
    }
    __device__ __forceinline__ int32_t gs() const { return (int32_t)(z & 0x7FFFFFFFu); }
    __device__ __forceinline__ int32_t ge() const { return (int32_t)(w & 0x7FFFFFFFu); }
    __device__ __forceinline__ uint32_t minus() const { return z >> 31; }
    __device__ __forceinline__ uint32_t dot() const { return w >> 31; }
};
template <> struct alignas(16) LiftIv<int64_t> {
    uint32_t x, y, fl, _pad;
    int64_t s, e;
    static __device__ __forceinline__ LiftIv make(uint32_t x, uint32_t y, int64_t gs, int64_t ge, uint32_t minus, uint32_t dot) {
        return LiftIv{x, y, minus | (dot << 1), 0u, gs, ge};
    }
    __device__ __forceinline__ int64_t gs() const { return s; }
    __device__ __forceinline__ int64_t ge() const { return e; }
    __device__ __forceinline__ uint32_t minus() const { return fl & 1u; }
    __device__ __forceinline__ uint32_t dot() const { return (fl >> 1) & 1u; }
};
static constexpr uint32_t LIFT_STRIP = 512 + 64; // slots whose owners are looked up from one scatter (bytes of LDS per wavefront)

// Pass 2, one record per lane, in rounds of up to 64 record slots: consecutive intervals share a round as long as their
// records fit.  sStrip maps record slots to owner lanes (owners mark their first slot — once per 512 slots, not per round —
// and a running maximum spreads the marks to the right); sIv / sOff hold the owners' interval data and first output line.
template <typename C>
__device__ __forceinline__ void lift_wave_emit(const ComposedRec<C> *__restrict__ recs, uint8_t *sStrip, const LiftIv<C> *sIv,
                                               const uint32_t *sOff, const int lane, const uint32_t b, const uint32_t p,
                                               const uint32_t totalSlots, const uint32_t firstQuery, const int64_t *__restrict__ tSeqStart,
                                               const int64_t ss0, const bool oneSeq, hgx_record *__restrict__ out,
                                               unsigned long long *sStage = nullptr) {
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t windowBase = 0;
    bool haveWindow = false;
    for (uint32_t base = 0; base < totalSlots;) {
        // intervals of this round: from the first one whose records start at `base` up to the first one that does not fit
        const bool started = p >= base;
        const bool fits = started && p + b <= base + 64;
        const unsigned long long stop = __ballot(started && !fits);
        const int qb = stop ? __ffsll((long long)stop) - 1 : 64;
        const uint32_t nextBase = qb < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)p, qb) : totalSlots;
        if (!haveWindow || base - windowBase > 512u) { // (wave-uniform; usually once per tile)
            windowBase = base;
            haveWindow = true;
            wave_lds_fence();
            reinterpret_cast<unsigned long long *>(sStrip)[lane] = 0ull;
            if (lane < 8)
                reinterpret_cast<unsigned long long *>(sStrip)[64 + lane] = 0ull;
            wave_lds_fence();
            if (b > 0 && started && p - base < LIFT_STRIP)
                sStrip[p - base] = (uint8_t)(lane + 1);
            wave_lds_fence();
        }
        const uint32_t mark = wave_incl_scan<LiftMax>((uint32_t)sStrip[base - windowBase + (uint32_t)lane]);
        const bool slotValid = base + (uint32_t)lane < nextBase && mark > 0;
        const int owner = mark > 0 ? (int)mark - 1 : 0;
        const LiftIv<C> iv = sIv[owner];
        const uint32_t oOff = sOff[owner];
        const C oGs = iv.gs(), oGe = iv.ge();
        const uint32_t lo = (iv.y - base) & 63u; // first slot of my interval in this round
        ComposedRec<C> r{};
        if (slotValid)
            r = recs[iv.x + base + (uint32_t)lane];
        const C pLo = r.sLo, pHi = r.sLo + r.len - 1;
        const bool emit = slotValid && pLo <= oGe && pHi >= oGs;
        const unsigned long long em = __ballot(emit);
        // lines of my interval before me (the table's order)
        const unsigned long long mine = em & ~((1ull << lo) - 1ull); // (records of earlier intervals sit below lo)
        uint32_t pos = (uint32_t)__popcll(mine & below);
        const C c = pLo > oGs ? pLo : oGs, d = pHi < oGe ? pHi : oGe;
        const C n = d - c + 1, delta = c - pLo;
        const uint32_t trev = r.mEncF & 1u;
        const C tLo = r.so + (trev ? r.len - delta - n : delta);
        // records that begin at or before the interval's first base all start there after clipping: among themselves
        // they go by target start.  They are the first lines of the interval; more than one only with paralogs.
        const bool inGroup = emit && pLo <= oGs;
        if (__any(inGroup && pos > 0)) {
            const int rel = (int)lane - (int)lo;
            const int span = (int)__builtin_amdgcn_readlane((int)wave_incl_scan<LiftMax>(inGroup ? (uint32_t)(rel + 1) : 0u), 63);
            const uint32_t oB = (uint32_t)__shfl((int)b, owner);
            uint32_t rank = 0;
            // (one exchange per partner: a lane that is not in such a group shows the largest coordinate, which is in front of nobody —
            // with paralogs an interval's first base lies in twenty records and this loop was most of the kernel)
            const C key = inGroup ? tLo : LiftCoord<C>::MAXV;
            // (four partners a trip, their exchanges asked for together: one after the other every partner cost an LDS round trip —
            // twenty of them a round on the 50-genome alignment, most of the kernel there)
            for (int jj = 0; jj < span; jj += 4) {
                const int p0 = (int)lo + jj, p1 = p0 + 1, p2 = p0 + 2, p3 = p0 + 3;
                const C t0 = wave_pull<C>(key, p0 & 63), t1 = wave_pull<C>(key, p1 & 63), t2 = wave_pull<C>(key, p2 & 63), t3 = wave_pull<C>(key, p3 & 63);
                if (inGroup) {
                    rank += ((uint32_t)jj < oB && p0 < 64 && t0 < tLo ? 1u : 0u) + ((uint32_t)(jj + 1) < oB && p1 < 64 && t1 < tLo ? 1u : 0u) +
                            ((uint32_t)(jj + 2) < oB && p2 < 64 && t2 < tLo ? 1u : 0u) + ((uint32_t)(jj + 3) < oB && p3 < 64 && t3 < tLo ? 1u : 0u);
                }
            }
            if (inGroup)
                pos = rank;
        }
        // (sStage — HGX_LIFT_STAGED: a round's records are one span of the output — it is dense in input order — unless a general interval's
        // lines lie between them: staged in LDS at their places and stored as the span it is, 8 bytes a lane and 512 a store,
        // instead of 40-byte records 40 bytes apart.  sOff[64]: where the wavefront's lines end)
        bool staged = false;
        uint32_t roundStart = 0;
        if (sStage) {
            const unsigned long long st = __ballot(started);
            const int qa = st ? __ffsll((long long)st) - 1 : 0;
            roundStart = sOff[qa];
            staged = em != 0 && sOff[qb] - roundStart == (uint32_t)__popcll(em);
        }
        if (emit) {
            const uint32_t seq = r.mEncF >> 8;
            const int64_t ss = oneSeq ? ss0 : tSeqStart[seq];
            const uint32_t rev = trev ^ iv.minus();
            if (staged) {
                unsigned long long *w = sStage + (size_t)(oOff + pos - roundStart) * 5;
                w[0] = (unsigned long long)(firstQuery + (uint32_t)owner);
                w[1] = (unsigned long long)((int64_t)tLo - ss);
                w[2] = (unsigned long long)((int64_t)tLo + n - ss);
                w[3] = (unsigned long long)(int64_t)c;
                w[4] = (unsigned long long)seq | ((unsigned long long)(uint8_t)(iv.dot() ? '.' : (rev ? '-' : '+')) << 32) | ((unsigned long long)rev << 40);
            } else {
            hgx_record rec;
            rec.query = (int64_t)(firstQuery + (uint32_t)owner);
            rec.tgt_start = (int64_t)tLo - ss;
            rec.tgt_end = (int64_t)tLo + n - ss;
            rec.src_start = (int64_t)c;
            rec.tgt_seq = (int32_t)seq;
            rec.strand = iv.dot() ? '.' : (rev ? '-' : '+');
            rec.tgt_reversed = (uint8_t)rev;
            rec._pad[0] = rec._pad[1] = 0;
            out[oOff + pos] = rec;
            }
        }
        if (staged) { // (wave-uniform)
            wave_lds_fence();
            const uint32_t words = (uint32_t)__popcll(em) * 5u;
            unsigned long long *dst = (unsigned long long *)(out + roundStart);
            for (uint32_t i = (uint32_t)lane; i < words; i += 64u)
                dst[i] = sStage[i];
            wave_lds_fence();
        }
        base = nextBase;
    }
}

// kb, nOut, waveTotal, groupTotal: k_lift_classify's and k_lift_totals' answers (waveExtra: the lines its workers counted, null
// in a run without them); genOffset /
// genRecords: where the general path left the records of the general intervals (nOut[q] records at genRecords +
// genOffset[q]); out / outCap: the dense output; outOffset[q]: first record of interval q in it.
template <typename C, int MINW>
static __global__ void __launch_bounds__(256, MINW) k_lift_merged(const int64_t *__restrict__ gStart, const int64_t *__restrict__ gEnd,
                                                            const uint8_t *__restrict__ strand, uint32_t nq, const uint2 *__restrict__ kb,
                                                            const ComposedRec<C> *__restrict__ recs, const int64_t *__restrict__ tSeqStart,
                                                            int tNumSeq, const uint32_t *__restrict__ genOffset,
                                                            const hgx_record *__restrict__ genRecords, hgx_record *__restrict__ out,
                                                            uint32_t outCap, const uint32_t *__restrict__ nOut, uint32_t *__restrict__ outOffset,
                                                            const uint32_t *__restrict__ waveTotal, const uint32_t *__restrict__ waveExtra,
                                                            const unsigned long long *__restrict__ groupTotal, uint32_t nTiles, int stagedStores) {
    __shared__ __attribute__((aligned(16))) uint8_t sStripAll[4][LIFT_STRIP];
    __shared__ LiftIv<C> sIvAll[4][64];
    __shared__ uint32_t sOffAll[4][65];
    __shared__ unsigned long long sStageAll[4][64 * 5];
    __shared__ uint32_t sWaveTotal[4], sFront[4];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const int64_t ss0 = tSeqStart[0];
    const bool oneSeq = tNumSeq <= 1;
    LIFT_PROF_DECL;
    for (uint32_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        // ---- the workgroup's intervals, one per thread ----
        const uint32_t q = tile * (uint32_t)LIFT_TILE + threadIdx.x;
        uint32_t k = 0, b = 0, flags = 0, cnt = 0, gOff = 0;
        C gs = 0, ge = -1;
        bool general = false;
        if (q < nq) {
            const int64_t s64 = gStart[q], e64 = gEnd[q];
            const uint8_t st = strand[q];
            const uint2 x = kb[q];
            cnt = nOut[q];
            gs = (C)s64;
            ge = (C)(e64 < (int64_t)LiftCoord<C>::MAXV ? e64 : (int64_t)LiftCoord<C>::MAXV);
            flags = (st == '-' ? 1u : 0u) | (st == '.' ? 2u : 0u);
            k = x.x;
            general = (x.y & KB_GENERAL) != 0;
            b = general ? 0u : x.y;
            if (general)
                gOff = genOffset[q];
        }
        // ---- the tile's place in the output ----
        // thread t looks at wavefront t of the tile's group: the ones in front of the tile, the ones of the tile in front of
        // this wavefront; the groups in front are added up by every wavefront for itself
        const uint32_t g = tile >> 6, j = tile & 63u;
        const uint32_t nWaves = (nq + 63u) >> 6;
        const uint32_t peer = (g << 8) + threadIdx.x;
        const uint32_t peerLines = peer < nWaves && threadIdx.x < 4u * j + 4u ? waveTotal[peer] + (waveExtra ? waveExtra[peer] : 0u) : 0u;
        unsigned long long before = 0;
        for (uint32_t g0 = 0; g0 < g; g0 += 64u)
            before += g0 + (uint32_t)lane < g ? groupTotal[g0 + (uint32_t)lane] : 0ull;
        const uint32_t inclSlots = wave_incl_scan<LiftSum>(b);
        const uint32_t p = inclSlots - b, totalSlots = (uint32_t)__builtin_amdgcn_readlane((int)inclSlots, 63);
        const uint32_t inclLines = wave_incl_scan<LiftSum>(cnt);
        sIvAll[w][lane] = LiftIv<C>::make(k - p, p, gs, ge, flags & 1u, (flags >> 1) & 1u);
        const uint32_t inFront = wave_total(threadIdx.x < 4u * j ? peerLines : 0u); // of the tiles in front, as far as this wavefront's threads see them
        if (lane == 0)
            sFront[w] = inFront;
        if (threadIdx.x >= 4u * j && threadIdx.x < 4u * j + 4u)
            sWaveTotal[threadIdx.x - 4u * j] = peerLines;
        const unsigned long long groupsBefore = wave_sum64(before);
        LIFT_PROF(0) // the intervals, their reach and their place have arrived
        __syncthreads();
        LIFT_PROF(1)
        const unsigned long long tileBase = groupsBefore + sFront[0] + sFront[1] + sFront[2] + sFront[3];
        uint32_t wavePrefix = 0, tileLines = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t = sWaveTotal[i];
            wavePrefix += i < w ? t : 0u;
            tileLines += t;
        }
        if (tileBase + tileLines <= (unsigned long long)outCap) { // (uniform over the workgroup; k_lift_totals has told the host otherwise)
            // ---- the lines, at their final place ----
            const uint32_t lineOff = (uint32_t)tileBase + wavePrefix + (inclLines - cnt);
            sOffAll[w][lane] = lineOff; // (made visible to the wave by the fences in front of lift_wave_emit's first scatter)
            if (lane == 63)
                sOffAll[w][64] = lineOff + cnt;
            lift_wave_emit(recs, sStripAll[w], sIvAll[w], sOffAll[w], lane, b, p, totalSlots, tile * (uint32_t)LIFT_TILE + (uint32_t)(w << 6),
                           tSeqStart, ss0, oneSeq, out, stagedStores ? sStageAll[w] : nullptr);
            LIFT_PROF(2) // lines stored
            // general intervals: their records were made by the general path; copied in as 8-byte words
            unsigned long long gm = __ballot(general && cnt > 0);
            while (gm) {
                const int o = __ffsll((long long)gm) - 1;
                gm &= gm - 1;
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)cnt, o), from = (uint32_t)__builtin_amdgcn_readlane((int)gOff, o),
                               to = (uint32_t)__builtin_amdgcn_readlane((int)lineOff, o);
                const unsigned long long *src = (const unsigned long long *)(genRecords + from);
                unsigned long long *dst = (unsigned long long *)(out + to);
                for (uint32_t i = (uint32_t)lane; i < n * 5u; i += 64u)
                    dst[i] = src[i];
            }
            if (q < nq)
                outOffset[q] = lineOff;
            LIFT_PROF(3)
        }
        if (tile + gridDim.x < nTiles)
            __syncthreads(); // (sWaveTotal, sFront are rewritten for the next tile)
    }
#if defined(HGX_LIFT_PROFILE) && HGX_LIFT_PROFILE == 1
    LIFT_PROF_FLUSH
#endif
}

// k_lift_totals, one workgroup between the counting and the storing launch:
//   groupTotal[g] = the lines of the 64 tiles (256 wavefronts' worth of intervals) of group g — after the finishing kernels have
//   added theirs to waveTotal, and with what k_lift_classify's workers counted in waveExtra —, *total = all lines, CNT_OVERFLOW if they do not fit the output (k_lift_merged then leaves
//   the tiles beyond it alone and the host repeats the batch with larger buffers);
//   folds the spread statistics words (hgx_liftover_kernels.hpp: stat_add) and writes everything the host wants to know of
//   the batch into LIFT_RB_WORDS words of host memory (rb is mapped pinned memory: no copy behind the last launch):
//   rb[0..7] = the scalar slots (CNT_MAPPED with the pieces of this run), rb[8 + l] = top-slot count of launch l (l < 4),
//   rb[12] = listed general intervals (HGX_FINISH_WAVE=0), rb[13] = the ones passed on to the finishing kernels,
//   rb[14] = bottom-slot count of launch 0 (general intervals);
//   leaves the words the next single-pass run counts in zeroed (the scalar slots, the level-0 append counters, the
//   statistics copies, the list counts): a batch then needs no memsets.  (k_lift_merged counts nothing.)
static constexpr int LIFT_RB_WORDS = 16;
static __global__ void __launch_bounds__(1024) k_lift_totals(const uint32_t *__restrict__ waveTotal, const unsigned long long *__restrict__ groupExtra,
                                                             uint32_t nWaves, uint32_t nGroups,
                                                             unsigned long long *__restrict__ groupTotal, uint32_t outCap,
                                                             uint32_t *__restrict__ total, unsigned long long *counters,
                                                             unsigned long long *generalCount, unsigned long long *restCount,
                                                             unsigned long long *workCounts, int storeLaunch, unsigned long long *rb) {
    __shared__ unsigned long long sums[7], sGrand[16], sFast;
    const int w = (int)threadIdx.x, lane = lane_id(), wv = w >> 6;
    unsigned long long grand = 0;
    // a wavefront per group, four groups' loads in flight at a time (the counts were written by other XCDs' workgroups: every
    // trip is one to the memory side)
    for (uint32_t g0 = (uint32_t)wv; g0 < nGroups; g0 += 64u) {
        unsigned long long part[4];
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            const uint32_t g = g0 + 16u * b;
            part[b] = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t idx = (g << 8) + (i << 6) + (uint32_t)lane;
                part[b] += g < nGroups && idx < nWaves ? (unsigned long long)waveTotal[idx] : 0ull;
            }
        }
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            const uint32_t g = g0 + 16u * b;
            // (+ the lines k_lift_classify's workers made for intervals of the group)
            const unsigned long long t = wave_sum64(part[b]) + (groupExtra && g < nGroups ? groupExtra[g] : 0ull);
            if (lane == 0 && g < nGroups)
                groupTotal[g] = t;
            grand += t;
        }
    }
    if (lane == 0)
        sGrand[wv] = grand;
    if (wv < 7) { // the copies of seven statistics words, a wavefront each
        const int word = wv == 0 ? STAT_MAPPED : wv == 5 ? STAT_LAUNCH0 + 1 : wv == 6 ? STAT_LAUNCH0 + 2 * storeLaunch + 1 : STAT_LAUNCH0 + 2 * (wv - 1);
        const unsigned long long sum = wave_sum64(lane < STAT_LINES ? counters[CNT_DSTAT0 + (size_t)lane * STAT_PITCH + word] : 0ull);
        if (lane == 0)
            sums[wv] = sum;
    }
    __syncthreads();
    if (w == 0) {
        unsigned long long all = 0;
        for (int i = 0; i < 16; ++i)
            all += sGrand[i];
        sFast = all - sums[6]; // lines from merged records = merged records that overlap their interval
        *total = (uint32_t)all;
        counters[CNT_LIFT_TOTAL] = all;
        if (all > (unsigned long long)outCap)
            counters[CNT_OVERFLOW] = 1;
    }
    __syncthreads();
    if (w < 8)
        rb[w] = counters[w] + (w == CNT_MAPPED ? sums[0] + sFast : 0ull);
    else if (w < 12)
        rb[w] = w - 8 == storeLaunch ? sFast : sums[w - 7];
    else if (w == 12)
        rb[w] = *generalCount;
    else if (w == 13)
        rb[w] = restCount ? *restCount : 0ull;
    else if (w == 14)
        rb[w] = sums[5];
    __syncthreads();
    if (w < 8)
        counters[w] = 0;
    if (w < NSEG)
        counters[CNT_FRONT0 + (size_t)w * SEG_PITCH] = 0;
    for (int i = w; i < STAT_LINES * (STAT_LAUNCH0 + 8); i += (int)blockDim.x)
        counters[CNT_DSTAT0 + (size_t)(i / (STAT_LAUNCH0 + 8)) * STAT_PITCH + (size_t)(i % (STAT_LAUNCH0 + 8))] = 0;
    if (w == 0 && restCount)
        *restCount = 0;
    if (w == 1)
        *generalCount = 0;
    if (w >= 64 && w < 64 + (int)LIFT_LISTS && workCounts)
        workCounts[(size_t)(w - 64) * LIFT_LIST_PITCH] = 0;
#ifdef HGX_LIFT_PROFILE
    __shared__ unsigned long long profSum[8];
    if (w < 8)
        profSum[w] = 0;
    __syncthreads();
    for (int i = w; i < 8192 * 4 * 8; i += (int)blockDim.x) {
        atomicAdd(&profSum[i & 7], g_liftProfile[i]);
        g_liftProfile[i] = 0;
    }
    __syncthreads();
    if (w == 0) {
        const double n = (double)profSum[7];
        // (both kernels write the same slots: the later one, k_lift_merged, unless built with -DHGX_LIFT_PROFILE=2, which leaves
        // it out: 1 = k_lift_merged: load+place, barrier, store, tail; 2 = k_lift_classify: classify, general, count, reduce,
        // then inside general_interval: table look-ups and records, finish_wave, reservation, store)
        printf("lift profile: waves %.0f, cycles per wave: %.0f %.0f %.0f %.0f | %.0f %.0f %.0f\n", n, (double)profSum[0] / n, (double)profSum[1] / n,
               (double)profSum[2] / n, (double)profSum[3] / n, (double)profSum[4] / n, (double)profSum[5] / n, (double)profSum[6] / n);
    }
#endif
}

} // namespace hgx
