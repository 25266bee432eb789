// The single-pass liftover kernels over the MERGED table (hgx_merged_kernels.hpp): BlockLiftover::liftInterval
// (liftover/impl/halBlockLiftover.cpp:46-113) for a batch of intervals — toSite, the per-source-segment halMapSegment calls,
// insertAndBreakOverlaps, extractSegment, the stable sort on the source start (liftover/impl/halLiftover.cpp:90) — as
//   k_lift_classify   one bucket look-up per interval end: where its records start, how many can be in reach, and whether
//                     the interval must go the general way (hgx_finish_kernel.hpp, over the unmerged table);
//   k_lift_merged     everything else: reads an interval's merged records, clips them, orders them, and writes the
//                     hgx_records of the whole batch densely and in input order — once.
// An unflagged interval's records have pairwise disjoint target ranges, so (see hgx_merged_kernels.hpp) its output lines
// are exactly its records clipped to it.  The reference prints them stably sorted by source start, ties in target order:
// the table is sorted by (source start, target start), so the clipped records already come in that order except for the
// ones that begin at or before the interval's first base — they all start at that base after clipping and are ordered by
// target start among themselves (they are the first of the interval's records; usually there is one).
//
// Dense output in input order from one pass needs every interval's offset = the number of lines of all intervals before
// it.  A workgroup owns a tile of 256 consecutive intervals; it counts its lines, publishes the count, obtains the sum of
// the tiles before it by a two-level decoupled look-back (64 tiles form a group; a tile reads the counts of the tiles
// before it in its own group and the totals of the groups before its own), then walks its records a second time (they are
// in the L1/L2 by then) and stores the lines at their final place.  The words the workgroups exchange are 8-byte
// {tag, value} granules written with one agent-scope store and polled with agent-scope loads (per-XCD L2s are not
// coherent; see cdna_hip_programming.md Guideline 16): no fences, nothing else is shared.  Tiles are dealt round-robin to a
// grid that is resident as a whole, so a tile only ever waits for workgroups that are running; every spin is bounded and a
// time-out makes the host repeat the batch on the multi-kernel path.
#pragma once
#include "../../include/hgx.h"
#include "hgx_liftover_kernels.hpp"

namespace hgx {

static constexpr int LIFT_TILE = 256;        // intervals per tile (= threads per workgroup)
static constexpr uint32_t LIFT_MAX_BOUND = 64; // records in reach of an interval the wave-wide rounds can hold
static constexpr uint32_t KB_GENERAL = 0x80000000u;
enum { CNT_LIFT_TOTAL = 1, CNT_LIFT_FAIL = 2 }; // counters[] slots (hgx_liftover_kernels.hpp uses 0 and 3..7)

// ---------------------------------------------------------------------------------------------
// k_lift_classify: kb[q] = {first record in reach, number of records in reach | KB_GENERAL}.  General intervals (a
// flagged record in reach, more than LIFT_MAX_BOUND records in reach, longer than the table's window) are listed per
// workgroup — a workgroup owns a contiguous range of intervals and a private slice of the list, so no atomics — and
// gathered into one dense list by k_lift_gather.
static __global__ void __launch_bounds__(256) k_lift_classify(const int64_t *__restrict__ gStart, const int64_t *__restrict__ gEnd, uint32_t nq,
                                                              int64_t genomeLength, const uint2 *__restrict__ coarseF,
                                                              const uint2 *__restrict__ startsF, int shift, int64_t window,
                                                              uint2 *__restrict__ kb, uint32_t *__restrict__ blockList,
                                                              uint32_t *__restrict__ blockCount, uint32_t chunk) {
    __shared__ uint32_t sCount;
    if (threadIdx.x == 0)
        sCount = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * chunk, hi = lo + chunk < nq ? lo + chunk : nq;
    for (uint32_t q = lo + threadIdx.x; q < hi; q += blockDim.x) {
        const int64_t gs = gStart[q], ge = gEnd[q];
        uint2 out = make_uint2(0u, 0u);
        if (ge >= gs && gs >= 0 && gs < genomeLength) {
            const int64_t geIn = ge < genomeLength ? ge : genomeLength - 1;
            const uint2 a = coarseF[gs >> shift], b = startsF[(geIn >> shift) + 1];
            if (b.x > a.x) {
                const uint32_t bound = b.x - a.x;
                const bool general = bound > LIFT_MAX_BOUND || b.y != a.y || ge - gs >= window;
                out = make_uint2(a.x, general ? KB_GENERAL : bound);
                if (general)
                    blockList[lo + atomicAdd(&sCount, 1u)] = q; // (LDS atomic; the slice holds the whole chunk)
            }
        }
        kb[q] = out;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        blockCount[blockIdx.x] = sCount;
}

// dense general list: every workgroup scans the (at most 2048) per-workgroup counts itself and copies its share
static __global__ void __launch_bounds__(256) k_lift_gather(const uint32_t *__restrict__ blockList, const uint32_t *__restrict__ blockCount,
                                                            uint32_t numBlocks, uint32_t chunk, uint32_t *__restrict__ list,
                                                            unsigned long long *__restrict__ listCount) {
    __shared__ uint32_t sPrefix[2049];
    __shared__ uint32_t sPart[256];
    const uint32_t per = (numBlocks + 255) / 256;
    uint32_t s = 0;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t b = threadIdx.x * per + k;
        s += b < numBlocks ? blockCount[b] : 0u;
    }
    sPart[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        uint32_t t = 0;
        if ((int)threadIdx.x >= o)
            t = sPart[threadIdx.x - o];
        __syncthreads();
        sPart[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t acc = sPart[threadIdx.x] - s;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t b = threadIdx.x * per + k;
        if (b < numBlocks) {
            sPrefix[b] = acc;
            acc += blockCount[b];
        }
    }
    if (threadIdx.x == 255)
        sPrefix[numBlocks] = sPart[255];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *listCount = (unsigned long long)sPrefix[numBlocks];
    for (uint32_t b = blockIdx.x; b < numBlocks; b += gridDim.x) {
        const uint32_t n = sPrefix[b + 1] - sPrefix[b];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
            list[sPrefix[b] + i] = blockList[(size_t)b * chunk + i];
    }
}

// ---------------------------------------------------------------------------------------------
// granules of the look-back: tag in the two top bits (0 = not there yet, 1 = the tile's / group's own count, 2 = count of
// everything up to and including it), value below
typedef __attribute__((address_space(1))) unsigned long long lift_gu64;
static constexpr unsigned long long LIFT_TAG_OWN = 1ull << 62, LIFT_TAG_INCL = 2ull << 62, LIFT_VALUE = (1ull << 62) - 1ull;
__device__ __forceinline__ void lift_publish(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store((lift_gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lift_peek(const unsigned long long *p) {
    return __hip_atomic_load((lift_gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static constexpr unsigned LIFT_SPIN_LIMIT = 1u << 17; // polls of one wait before the batch is given up (~0.1 s)
// a wait also ends when another workgroup has given up (looked at every 256 polls)
__device__ __forceinline__ bool lift_spin_over(unsigned spins, const unsigned long long *counters) {
    return spins >= LIFT_SPIN_LIMIT || ((spins & 255u) == 255u && lift_peek(&counters[CNT_LIFT_FAIL]) != 0);
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(v, o);
        if (lane >= o)
            v += up;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o);
    return __shfl(v, 0);
}

// One wavefront's pass over the records in reach of its 64 intervals, in rounds of up to 64 records: consecutive
// intervals share a round as long as their records fit.  EMIT = false counts every interval's lines, EMIT = true stores
// them.  Interval data live in the lane that owns the interval (lane = interval index inside the wave's 64) and reach the
// record lanes by ds_bpermute; sOwner is the wave's 64-byte LDS strip that maps a round's record slots to owner lanes.
template <bool EMIT>
__device__ __forceinline__ uint32_t lift_wave_pass(const ComposedRec<int32_t> *__restrict__ recs, uint8_t *sOwner, const int lane, const uint32_t k,
                                                   const uint32_t b, const uint32_t p, const uint32_t totalSlots, const int32_t gs,
                                                   const int32_t ge, const uint32_t flags /* bit 0 minus, bit 1 dot */, const uint32_t lineOff,
                                                   const uint32_t firstQuery, const int64_t *__restrict__ tSeqStart, const int64_t ss0,
                                                   const bool oneSeq, hgx_record *__restrict__ out, uint32_t &used) {
    uint32_t cnt = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t base = 0; base < totalSlots;) {
        // intervals of this round: from the first one whose records start at `base` up to the first one that does not fit
        const bool started = p >= base;
        const bool fits = started && p + b <= base + 64;
        const unsigned long long stop = __ballot(started && !fits);
        const int qb = stop ? __ffsll((long long)stop) - 1 : 64;
        const uint32_t nextBase = qb < 64 ? (uint32_t)__shfl((int)p, qb) : totalSlots;
        const bool inRound = fits && lane < qb && b > 0;
        // record slot -> owner lane: owners mark their first slot, a running maximum spreads the marks to the right
        sOwner[lane] = 0;
        wave_lds_fence();
        if (inRound)
            sOwner[p - base] = (uint8_t)(lane + 1);
        wave_lds_fence();
        int mark = (int)sOwner[lane];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(mark, o);
            if (lane >= o && up > mark)
                mark = up;
        }
        const bool slotValid = base + (uint32_t)lane < nextBase && mark > 0;
        const int owner = mark > 0 ? mark - 1 : 0;
        const uint32_t oK = (uint32_t)__shfl((int)k, owner), oP = (uint32_t)__shfl((int)p, owner);
        const int32_t oGs = __shfl(gs, owner), oGe = __shfl(ge, owner);
        const uint32_t lo = (oP - base) & 63u; // first slot of my interval in this round
        ComposedRec<int32_t> r{};
        if (slotValid)
            r = recs[oK + (base + (uint32_t)lane - oP)];
        const int32_t pLo = r.sLo, pHi = r.sLo + r.len - 1;
        const bool emit = slotValid && pLo <= oGe && pHi >= oGs;
        const unsigned long long em = __ballot(emit);
        if (!EMIT) {
            if (inRound) {
                const uint32_t a = p - base, e = a + b; // my slots [a, e)
                const unsigned long long mask = (e >= 64 ? ~0ull : ((1ull << e) - 1ull)) & ~((1ull << a) - 1ull);
                cnt += (uint32_t)__popcll(em & mask);
            }
        } else {
            used += emit ? 1u : 0u;
            // lines of my interval before me (the table's order)
            const unsigned long long mine = em & ~((1ull << lo) - 1ull); // (records of earlier intervals sit below lo)
            uint32_t pos = (uint32_t)__popcll(mine & below);
            const int32_t c = pLo > oGs ? pLo : oGs, d = pHi < oGe ? pHi : oGe;
            const int32_t n = d - c + 1, delta = c - pLo;
            const uint32_t trev = r.mEncF & 1u;
            const int32_t tLo = r.so + (trev ? r.len - delta - n : delta);
            // records that begin at or before the interval's first base all start there after clipping: among themselves
            // they go by target start.  They are the first lines of the interval; more than one only with paralogs.
            const bool inGroup = emit && pLo <= oGs;
            if (__any(inGroup && pos > 0)) {
                const int rel = (int)lane - (int)lo;
                int span = inGroup ? rel + 1 : 0; // slots of my interval up to and including me
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const int other = __shfl_xor(span, o);
                    span = other > span ? other : span;
                }
                const uint32_t oB = (uint32_t)__shfl((int)b, owner);
                uint32_t rank = 0;
                for (int jj = 0; jj < span; ++jj) {
                    const int partner = (int)lo + jj;
                    const int pGroup = __shfl((int)inGroup, partner & 63);
                    const int32_t pT = __shfl(tLo, partner & 63);
                    if (inGroup && (uint32_t)jj < oB && partner < 64 && pGroup && pT < tLo)
                        ++rank;
                }
                if (inGroup)
                    pos = rank;
            }
            const uint32_t oOff = (uint32_t)__shfl((int)lineOff, owner), oFl = (uint32_t)__shfl((int)flags, owner);
            if (emit) {
                const uint32_t seq = r.mEncF >> 8;
                const int64_t ss = oneSeq ? ss0 : tSeqStart[seq];
                const uint32_t rev = trev ^ (oFl & 1u);
                hgx_record rec;
                rec.query = (int64_t)(firstQuery + (uint32_t)owner);
                rec.tgt_start = (int64_t)tLo - ss;
                rec.tgt_end = (int64_t)tLo + n - ss;
                rec.src_start = (int64_t)c;
                rec.tgt_seq = (int32_t)seq;
                rec.strand = (oFl & 2u) ? '.' : (rev ? '-' : '+');
                rec.tgt_reversed = (uint8_t)rev;
                rec._pad[0] = rec._pad[1] = 0;
                out[oOff + pos] = rec;
            }
        }
        base = nextBase;
    }
    return cnt;
}

// kb: k_lift_classify's answer; genOffset / genRecords / nOut: where the general path left the records of the general
// intervals (k_finish_lds: nOut[q] records at genRecords + genOffset[q]); out / outCap: the dense output; outOffset[q]:
// first record of interval q in it.  tileStatus / groupStatus: zeroed look-back granules.
template <int MINW>
static __global__ void __launch_bounds__(256, MINW) k_lift_merged(const int64_t *__restrict__ gStart, const int64_t *__restrict__ gEnd,
                                                            const uint8_t *__restrict__ strand, uint32_t nq, const uint2 *__restrict__ kb,
                                                            const ComposedRec<int32_t> *__restrict__ recs, const int64_t *__restrict__ tSeqStart,
                                                            int tNumSeq, const uint32_t *__restrict__ genOffset,
                                                            const hgx_record *__restrict__ genRecords, hgx_record *__restrict__ out,
                                                            uint32_t outCap, uint32_t *__restrict__ nOut, uint32_t *__restrict__ outOffset,
                                                            unsigned long long *tileStatus, unsigned long long *groupStatus, uint32_t nTiles,
                                                            unsigned long long *counters, unsigned long long *kstat, uint32_t *__restrict__ total) {
    __shared__ uint8_t sOwnerAll[4][64];
    __shared__ uint32_t sWaveTotal[4];
    __shared__ unsigned long long sTileBase;
    __shared__ int sGiveUp;
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    uint8_t *sOwner = sOwnerAll[w];
    const int64_t ss0 = tSeqStart[0];
    const bool oneSeq = tNumSeq <= 1;
    uint32_t used = 0;
    for (uint32_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        // ---- the workgroup's intervals, one per thread ----
        const uint32_t q = tile * (uint32_t)LIFT_TILE + threadIdx.x;
        uint32_t k = 0, b = 0, flags = 0, cnt = 0, gOff = 0;
        int32_t gs = 0, ge = -1;
        bool general = false;
        if (q < nq) {
            const uint2 x = kb[q];
            k = x.x;
            general = (x.y & KB_GENERAL) != 0;
            b = general ? 0u : x.y;
            const int64_t s64 = gStart[q], e64 = gEnd[q];
            gs = (int32_t)s64;
            ge = (int32_t)(e64 < 0x7FFFFFFFll ? e64 : 0x7FFFFFFFll);
            const uint8_t st = strand[q];
            flags = (st == '-' ? 1u : 0u) | (st == '.' ? 2u : 0u);
            if (general) {
                cnt = nOut[q];
                gOff = genOffset[q];
            }
        }
        const uint32_t inclSlots = wave_incl_scan(b, lane);
        const uint32_t p = inclSlots - b, totalSlots = (uint32_t)__shfl((int)inclSlots, 63);
        // ---- pass 1: lines per interval ----
        uint32_t dummy = 0;
        const uint32_t fast = lift_wave_pass<false>(recs, sOwner, lane, k, b, p, totalSlots, gs, ge, flags, 0u, 0u, tSeqStart, ss0, oneSeq, out,
                                                     dummy);
        if (!general)
            cnt = fast;
        const uint32_t inclLines = wave_incl_scan(cnt, lane);
        if (lane == 63)
            sWaveTotal[w] = inclLines;
        __syncthreads();
        uint32_t wavePrefix = 0, tileTotal = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t = sWaveTotal[i];
            wavePrefix += i < w ? t : 0u;
            tileTotal += t;
        }
        // ---- the tile's place in the output: two-level look-back by the first wavefront ----
        if (w == 0) {
            bool giveUp = false;
            if (lane == 0)
                lift_publish(&tileStatus[tile], LIFT_TAG_OWN | (unsigned long long)tileTotal);
            const uint32_t g = tile >> 6, j = tile & 63u;
            const uint32_t groupSize = nTiles - (g << 6) < 64u ? nTiles - (g << 6) : 64u;
            // (a) the tiles before mine in my group
            unsigned long long mine = 0;
            for (unsigned spins = 0;; ++spins) {
                const unsigned long long v = (uint32_t)lane < j ? lift_peek(&tileStatus[(g << 6) + (uint32_t)lane]) : LIFT_TAG_OWN;
                if (__all((v & ~LIFT_VALUE) != 0)) {
                    mine = wave_sum64((uint32_t)lane < j ? (v & LIFT_VALUE) : 0ull);
                    break;
                }
                if (lift_spin_over(spins, counters)) {
                    giveUp = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            const bool lastOfGroup = j + 1 == groupSize;
            if (lastOfGroup && lane == 0 && !giveUp)
                lift_publish(&groupStatus[g], LIFT_TAG_OWN | (mine + tileTotal));
            // (b) the groups before mine: 64 per step, stopping at the first one that knows its inclusive count
            unsigned long long before = 0;
            for (long long posn = (long long)g - 1; posn >= 0 && !giveUp;) {
                const long long idx = posn - lane;
                unsigned long long v = LIFT_TAG_INCL; // in front of group 0: inclusive count 0
                bool done = false;
                for (unsigned spins = 0;; ++spins) {
                    v = idx >= 0 ? lift_peek(&groupStatus[idx]) : LIFT_TAG_INCL;
                    const unsigned long long empty = __ballot((v & ~LIFT_VALUE) == 0);
                    const unsigned long long incl = __ballot((v & ~LIFT_VALUE) == LIFT_TAG_INCL);
                    // usable once every lane up to the nearest inclusive one (or all 64) has something
                    const int firstIncl = incl ? __ffsll((long long)incl) - 1 : 63;
                    const unsigned long long need = firstIncl >= 63 ? ~0ull : ((2ull << firstIncl) - 1ull);
                    if ((empty & need) == 0) {
                        before += wave_sum64(lane <= firstIncl ? (v & LIFT_VALUE) : 0ull);
                        done = incl != 0;
                        break;
                    }
                    if (lift_spin_over(spins, counters)) {
                        giveUp = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (done || giveUp)
                    break;
                posn -= 64;
            }
            if (lastOfGroup && lane == 0 && !giveUp)
                lift_publish(&groupStatus[g], LIFT_TAG_INCL | (before + mine + tileTotal));
            const unsigned long long tileBase = before + mine;
            int state = giveUp ? 1 : 0; // 1: a wait timed out (the host repeats the batch on the multi-kernel path)
            if (!giveUp && tileBase + tileTotal > (unsigned long long)outCap) { // the retry sizes the buffers from CNT_LIFT_TOTAL
                state = 2;
                if (lane == 0)
                    counters[CNT_OVERFLOW] = 1;
            }
            if (lane == 0) {
                sTileBase = tileBase;
                sGiveUp = state;
                if (tile + 1 == nTiles) {
                    *total = (uint32_t)(tileBase + tileTotal);
                    counters[CNT_LIFT_TOTAL] = tileBase + tileTotal;
                }
            }
        }
        __syncthreads();
        const int state = sGiveUp;
        const uint32_t tileBase = (uint32_t)sTileBase;
        if (state != 0) {
            if (threadIdx.x == 0 && state == 1)
                lift_publish(&counters[CNT_LIFT_FAIL], 1ull);
        } else {
            // ---- pass 2: the lines, at their final place ----
            const uint32_t lineOff = tileBase + wavePrefix + (inclLines - cnt);
            lift_wave_pass<true>(recs, sOwner, lane, k, b, p, totalSlots, gs, ge, flags, lineOff, tile * (uint32_t)LIFT_TILE + (uint32_t)(w << 6),
                                 tSeqStart, ss0, oneSeq, out, used);
            // general intervals: their records were made by the general path; copied in as 8-byte words
            unsigned long long gm = __ballot(general && cnt > 0);
            while (gm) {
                const int o = __ffsll((long long)gm) - 1;
                gm &= gm - 1;
                const uint32_t n = (uint32_t)__shfl((int)cnt, o), from = (uint32_t)__shfl((int)gOff, o), to = (uint32_t)__shfl((int)lineOff, o);
                const unsigned long long *src = (const unsigned long long *)(genRecords + from);
                unsigned long long *dst = (unsigned long long *)(out + to);
                for (uint32_t i = (uint32_t)lane; i < n * 5u; i += 64u)
                    dst[i] = src[i];
            }
            if (q < nq) {
                nOut[q] = cnt;
                outOffset[q] = lineOff;
            }
        }
        __syncthreads(); // (sWaveTotal, sTileBase are reused by the next tile)
    }
    stat_add(&kstat[0], used); // the "top" slot of this launch: merged records that overlap their interval
    stat_add(&counters[CNT_DSTAT0 + STAT_MAPPED], used);
}

// End of a single-pass run: folds the spread statistics words (hgx_liftover_kernels.hpp: stat_add) and gathers everything
// the host reads back into LIFT_RB_WORDS consecutive words — one 128-byte copy per batch.
// rb[0..7] = the scalar slots (CNT_MAPPED with the pieces of this run), rb[8 + l] = top-slot count of launch l (l < 4),
// rb[12] = general intervals, rb[13] = the ones k_general_wave passed on.
static constexpr int LIFT_RB_WORDS = 16;
// ... and leaves the words the next single-pass run counts in zeroed (the scalar slots, the level-0 append counters, the
// statistics copies, the look-back granules of this run, the count of passed-on intervals): a batch then needs no memsets.
static __global__ void __launch_bounds__(256) k_lift_epilogue(unsigned long long *counters, const unsigned long long *generalCount,
                                                              unsigned long long *restCount, unsigned long long *rb, unsigned long long *granules,
                                                              uint32_t numGranules) {
    __shared__ unsigned long long sums[5];
    const int w = (int)threadIdx.x;
    if (w < 5) {
        const int word = w == 0 ? STAT_MAPPED : STAT_LAUNCH0 + 2 * (w - 1);
        unsigned long long sum = 0;
        for (int l = 0; l < STAT_LINES; ++l)
            sum += counters[CNT_DSTAT0 + (size_t)l * STAT_PITCH + word];
        sums[w] = sum;
    }
    __syncthreads();
    if (w < 8)
        rb[w] = counters[w] + (w == CNT_MAPPED ? sums[0] : 0ull);
    else if (w < 12)
        rb[w] = sums[w - 7];
    else if (w == 12)
        rb[w] = *generalCount;
    else if (w == 13)
        rb[w] = restCount ? *restCount : 0ull; // intervals k_general_wave passed on
    __syncthreads();
    if (w < 8)
        counters[w] = 0;
    if (w < NSEG)
        counters[CNT_FRONT0 + (size_t)w * SEG_PITCH] = 0;
    for (int i = w; i < STAT_LINES * (STAT_LAUNCH0 + 8); i += (int)blockDim.x)
        counters[CNT_DSTAT0 + (size_t)(i / (STAT_LAUNCH0 + 8)) * STAT_PITCH + (size_t)(i % (STAT_LAUNCH0 + 8))] = 0;
    for (uint32_t i = (uint32_t)w; i < numGranules; i += blockDim.x)
        granules[i] = 0;
    if (w == 0 && restCount)
        *restCount = 0;
}

} // namespace hgx
