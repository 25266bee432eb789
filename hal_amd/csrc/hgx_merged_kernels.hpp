// Kernels that turn the table of the whole path (ComposedRec pieces of every source top segment, hgx_table_kernels.hpp)
// into the MERGED table the single-pass lift kernel reads (hgx_lift_kernels.hpp).
//
// Which two pieces of an interval end up on one output line does not depend on the interval: canMergeRightWith
// (api/impl/halMappedSegment.cpp:109-161) asks for equal orientations and for the two pieces to be neighbours by exactly
// one base on the source AND on the target, and a junction that lies inside an interval is never moved by clipping the
// pieces to the interval (only the outer ends at the interval's first and last base are).  So, as long as no two pieces of
// the interval overlap or tie on the target (then every equivalence class of BlockMapper::extractSegment,
// liftover/impl/halBlockMapper.cpp:331-394, has one member and no cut point exists), an interval's lines are the maximal
// chains of such neighbours clipped to the interval — and the chains can be built once, per table.  A merged record is one
// chain: (source start, length, forward target start, target strand, target sequence), a row of a chain file.
//
// The intervals that may need the general algorithm (overlap breaking, equivalence classes, cut points) are found without
// looking at their pieces: a merged record is FLAGGED when its target range overlaps the target range of another record
// whose source range lies within `window` bases of its own.  Two records that both touch an interval no longer than
// `window` are that close, so an interval shorter than the window whose records carry no flag has pairwise disjoint target
// ranges; everything else (a flagged record among the ones that overlap it, more than 64 records, a longer interval) goes
// the general way over the unmerged table.  The flag rides in the record (bit 1 of mEncF), so the classification costs
// nothing beyond the walk over the interval's records that counts its lines.
//
// Every kernel is a template on the coordinate type C of the alignment's tables (hal_index_t is int64, api/inc/halDefs.h:34;
// int32 tables serve alignments whose genomes are all shorter than 2^31 bases).  Only the sort keys differ: with 32-bit
// coordinates a junction (target position, source position, strand, side) packs into one 64-bit radix-sort key, with 64-bit
// coordinates it is sorted in two stable passes (source part first, target part second).
#pragma once
#include "hgx_liftover_kernels.hpp"

namespace hgx {

__device__ __forceinline__ int seq_of(const int64_t *__restrict__ seqStart, int numSeq, int64_t pos) {
    int lo = 0, hi = numSeq; // seqStart[lo] <= pos < seqStart[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seqStart[mid] <= pos)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Junctions.  A piece's RIGHT junction is the position just past its high target end together with the source position a
// mergeable right neighbour must start (forward target) or end (reversed target) at; its LEFT junction is its own low target
// end described the same way.  A merges with B on its right exactly when right(A) == left(B).
// v = piece index << 1 | (0 right, 1 left)
struct Junction {
    uint64_t t, s; // target junction; source junction << 2 | target strand << 1 | side
};
template <typename C> __device__ __forceinline__ Junction junction_of(const ComposedRec<C> &r, uint32_t side) {
    const uint64_t sLo = (uint64_t)r.sLo, sEnd = (uint64_t)r.sLo + (uint64_t)r.len; // one past the high source end
    const uint64_t tLo = (uint64_t)r.so, tEnd = (uint64_t)r.so + (uint64_t)r.len;
    const uint64_t trev = r.mEncF & 1u;
    // forward target: source and target run the same way, B continues where A's source ends; reversed target: B's source
    // ends where A's begins (halMappedSegment.cpp:131-150 in forward coordinates)
    const uint64_t rightSrc = trev ? sLo : sEnd, leftSrc = trev ? sEnd : sLo;
    Junction j;
    j.t = side ? tLo : tEnd;
    j.s = ((side ? leftSrc : rightSrc) << 2) | (trev << 1) | (uint64_t)side;
    return j;
}

// 32-bit tables: key = target junction << (sBits + 2) | source junction << 2 | target strand << 1 | side
// (sBits = bits of the source genome's length: the sort then runs over tBits + sBits + 2 key bits instead of 64)
static __global__ void __launch_bounds__(256) k_merge_keys(const ComposedRec<int32_t> *__restrict__ recs, uint32_t n, int sBits,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const ComposedRec<int32_t> r = recs[i];
    const Junction a = junction_of(r, 0u), b = junction_of(r, 1u);
    keys[2 * (size_t)i] = (a.t << (sBits + 2)) | a.s;
    vals[2 * (size_t)i] = 2u * i;
    keys[2 * (size_t)i + 1] = (b.t << (sBits + 2)) | b.s;
    vals[2 * (size_t)i + 1] = 2u * i + 1u;
}
// 64-bit tables, first pass: the source part of both junctions of every piece
static __global__ void __launch_bounds__(256) k_merge_keys_low(const ComposedRec<int64_t> *__restrict__ recs, uint32_t n, uint64_t *__restrict__ keys,
                                                               uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const ComposedRec<int64_t> r = recs[i];
    keys[2 * (size_t)i] = junction_of(r, 0u).s;
    vals[2 * (size_t)i] = 2u * i;
    keys[2 * (size_t)i + 1] = junction_of(r, 1u).s;
    vals[2 * (size_t)i + 1] = 2u * i + 1u;
}
// second pass: the target part, in the order the first pass left (the radix sort is stable)
static __global__ void __launch_bounds__(256) k_merge_keys_high(const ComposedRec<int64_t> *__restrict__ recs, const uint32_t *__restrict__ sortedVals,
                                                                uint32_t n2, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n2)
        return;
    const uint32_t v = sortedVals[p];
    keys[p] = junction_of(recs[v >> 1], v & 1u).t;
    vals[p] = v;
}

// After the sort a junction with exactly one right end followed by exactly one left end links two pieces — if they lie on
// the same target sequence (halBlockMapper.cpp:366).  Junctions claimed by more pieces are
// left alone: those pieces tie or overlap on the target at distance zero, so they are flagged and never read as merged.
// (The junctions are read back from the records: the sorted order is all the sort is needed for.)
template <typename C>
static __global__ void __launch_bounds__(256) k_merge_link(const uint32_t *__restrict__ vals, uint32_t n2, const ComposedRec<C> *__restrict__ recs,
                                                           const int64_t *__restrict__ tSeqStart, int tNumSeq, const int64_t *__restrict__ sSeqStart,
                                                           int sNumSeq, uint32_t *__restrict__ root) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + 1 >= n2)
        return;
    const uint32_t va = vals[p], vb = vals[p + 1];
    if ((va & 1u) != 0 || (vb & 1u) != 1)
        return;
    const uint32_t A = va >> 1, B = vb >> 1;
    if (A == B)
        return;
    const ComposedRec<C> ra = recs[A], rb = recs[B];
    const Junction a = junction_of(ra, 0u), b = junction_of(rb, 1u);
    if (a.t != b.t || (a.s >> 1) != (b.s >> 1))
        return;
    if (p > 0) {
        const uint32_t v = vals[p - 1];
        const Junction x = junction_of(recs[v >> 1], v & 1u);
        if (x.t == a.t && (x.s >> 1) == (a.s >> 1))
            return;
    }
    if (p + 2 < n2) {
        const uint32_t v = vals[p + 2];
        const Junction x = junction_of(recs[v >> 1], v & 1u);
        if (x.t == a.t && (x.s >> 1) == (a.s >> 1))
            return;
    }
    if (tNumSeq > 1 && seq_of(tSeqStart, tNumSeq, (int64_t)ra.so) != seq_of(tSeqStart, tNumSeq, (int64_t)rb.so))
        return;
    // (the source sequence is not looked at: canMergeRightWith only asserts that it is the same, halMappedSegment.cpp:118, and an
    // interval that reaches into the sequence in front — a negative chromStart, which the reference does not check — is merged
    // across the boundary there; intervals inside one sequence never see such a junction)
    (void)sSeqStart;
    (void)sNumSeq;
    root[B] = A; // B hangs under its left neighbour (root[] starts as the identity)
}

static __global__ void __launch_bounds__(256) k_merge_identity(uint32_t *__restrict__ root, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        root[i] = i;
}

// pointer jumping: after ceil(log2(longest chain)) rounds root[i] is the chain's leftmost (lowest target) piece
static __global__ void __launch_bounds__(256) k_merge_jump(uint32_t *__restrict__ root, uint32_t n, unsigned int *__restrict__ changed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t r = root[i];
    const uint32_t rr = root[r];
    if (rr != r) {
        root[i] = rr;
        *changed = 1;
    }
}

// the unsigned word a chain's extent is accumulated in
template <typename C> struct MergeWord;
template <> struct MergeWord<int32_t> {
    typedef uint32_t U;
};
template <> struct MergeWord<int64_t> {
    typedef unsigned long long U;
};

// a chain's extent, accumulated on its root: lowest source position, total length (the lowest target position is the root's own)
template <typename C>
static __global__ void __launch_bounds__(256) k_merge_extent(const ComposedRec<C> *__restrict__ recs, const uint32_t *__restrict__ root, uint32_t n,
                                                             typename MergeWord<C>::U *__restrict__ minS, typename MergeWord<C>::U *__restrict__ sumLen) {
    typedef typename MergeWord<C>::U U;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t r = root[i];
    atomicMin(&minS[r], (U)recs[i].sLo);
    atomicAdd(&sumLen[r], (U)recs[i].len);
}

// sort keys of the chains: (source start, target start); pieces that are not a chain's root sort behind everything.
// 32-bit tables: one key, source start << 32 | target start
static __global__ void __launch_bounds__(256) k_merge_heads(const ComposedRec<int32_t> *__restrict__ recs, const uint32_t *__restrict__ root, uint32_t n,
                                                            const uint32_t *__restrict__ minS, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                            unsigned int *__restrict__ numHeads) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool head = i < n && root[i] == i;
    if (i < n) {
        keys[i] = head ? (((uint64_t)minS[i] << 32) | (uint64_t)(uint32_t)recs[i].so) : ~0ull;
        vals[i] = i;
    }
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(numHeads, (unsigned int)__popcll(m));
}
// 64-bit tables, first pass: the target start
static __global__ void __launch_bounds__(256) k_merge_heads_low(const ComposedRec<int64_t> *__restrict__ recs, const uint32_t *__restrict__ root, uint32_t n,
                                                                uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                                unsigned int *__restrict__ numHeads) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool head = i < n && root[i] == i;
    if (i < n) {
        keys[i] = head ? (uint64_t)recs[i].so : ~0ull;
        vals[i] = i;
    }
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(numHeads, (unsigned int)__popcll(m));
}
// second pass: the source start
static __global__ void __launch_bounds__(256) k_merge_heads_high(const uint32_t *__restrict__ root, const uint32_t *__restrict__ sortedVals, uint32_t n,
                                                                 const unsigned long long *__restrict__ minS, uint64_t *__restrict__ keys,
                                                                 uint32_t *__restrict__ vals) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
        return;
    const uint32_t i = sortedVals[p];
    keys[p] = root[i] == i ? (uint64_t)minS[i] : ~0ull;
    vals[p] = i;
}

// merged records in (source start, target start) order.  mEncF = target strand | target sequence << 8
template <typename C>
static __global__ void __launch_bounds__(256) k_merge_records(const ComposedRec<C> *__restrict__ recs, const uint32_t *__restrict__ sortedHeads, uint32_t m,
                                                              const typename MergeWord<C>::U *__restrict__ minS,
                                                              const typename MergeWord<C>::U *__restrict__ sumLen, const int64_t *__restrict__ tSeqStart,
                                                              int tNumSeq, ComposedRec<C> *__restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m)
        return;
    const uint32_t h = sortedHeads[j];
    const ComposedRec<C> r = recs[h];
    ComposedRec<C> o{};
    o.sLo = (C)minS[h];
    o.len = (C)sumLen[h];
    o.so = r.so;
    const uint32_t seq = tNumSeq > 1 ? (uint32_t)seq_of(tSeqStart, tNumSeq, (int64_t)r.so) : 0u;
    o.mEncF = (r.mEncF & 1u) | (seq << 8);
    out[j] = o;
}

// ---- flags: target overlap with a record whose source lies within `window` bases ----
// sort key: the target start (payload: the record index)
template <typename C>
static __global__ void __launch_bounds__(256) k_flag_keys(const ComposedRec<C> *__restrict__ recs, uint32_t m, uint64_t *__restrict__ keys,
                                                          uint32_t *__restrict__ vals) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) {
        keys[j] = (uint64_t)recs[j].so;
        vals[j] = j;
    }
}
// high target ends in target-start order (input of the running maximum)
template <typename C>
static __global__ void __launch_bounds__(256) k_flag_ends(const ComposedRec<C> *__restrict__ recs, const uint32_t *__restrict__ sortedIdx, uint32_t m,
                                                          typename MergeWord<C>::U *__restrict__ tHiSorted) {
    typedef typename MergeWord<C>::U U;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m)
        return;
    const ComposedRec<C> x = recs[sortedIdx[r]];
    tHiSorted[r] = (U)x.so + (U)x.len - (U)1;
}
struct MaxOp {
    template <typename T> __host__ __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
};
static constexpr int FLAG_SCAN_CUT = 128; // neighbours looked at in either direction before giving up (and flagging)
// Every record looks at the records after it in target-start order while they begin inside its target range, and at the
// records before it while any of them can still reach it (running maximum of the high ends).  A pair is seen from its
// earlier member unless that scan was cut off — then the earlier member is flagged as it stands and the later member's
// backward scan either finds the pair or is cut off and flags itself.
template <typename C>
static __global__ void __launch_bounds__(256) k_flag_overlaps(const ComposedRec<C> *__restrict__ recs, const uint64_t *__restrict__ sortedTLo,
                                                              const uint32_t *__restrict__ sortedIdx, const typename MergeWord<C>::U *__restrict__ runMax,
                                                              uint32_t m, int64_t window, uint32_t *__restrict__ flag) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m)
        return;
    const uint32_t i = sortedIdx[r];
    const ComposedRec<C> x = recs[i];
    const int64_t tLo = (int64_t)x.so, tHi = tLo + (int64_t)x.len - 1, sLo = (int64_t)x.sLo, sHi = sLo + (int64_t)x.len - 1;
    auto near = [&](const ComposedRec<C> &y) {
        const int64_t yLo = (int64_t)y.sLo, yHi = yLo + (int64_t)y.len - 1;
        const int64_t gap = (sLo > yLo ? sLo : yLo) - (sHi < yHi ? sHi : yHi);
        return gap < window;
    };
    bool mine = false;
    int steps = 0;
    for (uint32_t r2 = r + 1; r2 < m; ++r2) {
        if ((int64_t)sortedTLo[r2] > tHi)
            break;
        if (++steps > FLAG_SCAN_CUT) {
            mine = true;
            break;
        }
        const uint32_t j = sortedIdx[r2];
        if (near(recs[j])) {
            mine = true;
            flag[j] = 1u;
        }
    }
    steps = 0;
    for (uint32_t r2 = r; r2-- > 0;) {
        if ((int64_t)runMax[r2] < tLo)
            break;
        if (++steps > FLAG_SCAN_CUT) {
            mine = true;
            break;
        }
        const uint32_t j = sortedIdx[r2];
        const ComposedRec<C> y = recs[j];
        if ((int64_t)y.so + (int64_t)y.len - 1 >= tLo && near(y)) {
            mine = true;
            flag[j] = 1u;
        }
    }
    if (mine)
        flag[i] = 1u;
}

// the flag rides in bit 1 of the record's mEncF; LIFT_SENTINELS records that begin behind every base end the table (the
// scans of k_lift_classify stop at the first record that begins behind their interval and read four records at a time)
static constexpr uint32_t LIFT_SENTINELS = 8;
template <typename C> struct LiftCoord;
template <> struct LiftCoord<int32_t> {
    static constexpr int32_t MAXV = 0x7FFFFFFF;
};
template <> struct LiftCoord<int64_t> {
    static constexpr int64_t MAXV = 0x7FFFFFFFFFFFFFFFll;
};
template <typename C>
static __global__ void __launch_bounds__(256) k_merge_mark(ComposedRec<C> *__restrict__ recs, const uint32_t *__restrict__ flag, uint32_t m) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) {
        if (flag[j])
            recs[j].mEncF |= 2u;
    } else if (j < m + LIFT_SENTINELS) {
        ComposedRec<C> o{};
        o.sLo = LiftCoord<C>::MAXV;
        o.len = 0;
        o.so = 0;
        o.mEncF = 0;
        recs[j] = o;
    }
}

// ---- which buckets a flagged record touches: k_lift_general_list finds the intervals that must go the general way from these
// bits alone (hgx_lift_kernels.hpp) ----
// bits[b >> 5] bit (b & 31): a flagged record touches bucket b (two words of slack behind the last bucket's)
template <typename C>
static __global__ void __launch_bounds__(256) k_bucket_flag_bits(const ComposedRec<C> *__restrict__ recs, const uint32_t *__restrict__ flag,
                                                                 uint32_t m, int shift, uint32_t *__restrict__ bits) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m || !flag[j])
        return;
    const int64_t lo = (int64_t)recs[j].sLo, hi = lo + (int64_t)recs[j].len - 1;
    for (int64_t b = lo >> shift; b <= (hi >> shift); ++b)
        atomicOr(&bits[b >> 5], 1u << (b & 31));
}
} // namespace hgx
