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
// 32-bit coordinates only (every genome < 2^31 bases); wider alignments keep to the unmerged table.
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

// Junction keys.  A piece's RIGHT key describes the position just past its high target end together with the source
// position a mergeable right neighbour must start (forward target) or end (reversed target) at; its LEFT key describes its
// own low target end the same way.  A merges with B on its right exactly when right(A) == left(B).
// key = target junction << 33 | source junction << 2 | target strand << 1 | (0 right, 1 left)
static __global__ void __launch_bounds__(256) k_merge_keys(const ComposedRec<int32_t> *__restrict__ recs, uint32_t n, uint64_t *__restrict__ keys,
                                                           uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const ComposedRec<int32_t> r = recs[i];
    const uint64_t sLo = (uint64_t)r.sLo, sEnd = (uint64_t)r.sLo + (uint64_t)r.len; // one past the high source end
    const uint64_t tLo = (uint64_t)r.so, tEnd = (uint64_t)r.so + (uint64_t)r.len;
    const uint64_t trev = r.mEncF & 1u;
    // forward target: source and target run the same way, B continues where A's source ends; reversed target: B's source
    // ends where A's begins (halMappedSegment.cpp:131-150 in forward coordinates)
    const uint64_t rightSrc = trev ? sLo : sEnd, leftSrc = trev ? sEnd : sLo;
    keys[2 * (size_t)i] = (tEnd << 33) | (rightSrc << 2) | (trev << 1) | 0u;
    vals[2 * (size_t)i] = i;
    keys[2 * (size_t)i + 1] = (tLo << 33) | (leftSrc << 2) | (trev << 1) | 1u;
    vals[2 * (size_t)i + 1] = i;
}

// After the sort a junction with exactly one right key followed by exactly one left key links two pieces — if they lie on
// the same target sequence (halBlockMapper.cpp:366) and the same source sequence.  Junctions claimed by more pieces are
// left alone: those pieces tie or overlap on the target at distance zero, so they are flagged and never read as merged.
static __global__ void __launch_bounds__(256) k_merge_link(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint32_t n2,
                                                           const ComposedRec<int32_t> *__restrict__ recs, const int64_t *__restrict__ tSeqStart,
                                                           int tNumSeq, const int64_t *__restrict__ sSeqStart, int sNumSeq,
                                                           uint32_t *__restrict__ root) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + 1 >= n2)
        return;
    const uint64_t a = keys[p], b = keys[p + 1];
    if ((a & 1u) != 0 || (b & 1u) != 1 || (a >> 1) != (b >> 1))
        return;
    if (p > 0 && (keys[p - 1] >> 1) == (a >> 1))
        return;
    if (p + 2 < n2 && (keys[p + 2] >> 1) == (a >> 1))
        return;
    const uint32_t A = vals[p], B = vals[p + 1];
    if (A == B)
        return;
    const ComposedRec<int32_t> ra = recs[A], rb = recs[B];
    if (tNumSeq > 1 && seq_of(tSeqStart, tNumSeq, (int64_t)ra.so) != seq_of(tSeqStart, tNumSeq, (int64_t)rb.so))
        return;
    if (sNumSeq > 1 && seq_of(sSeqStart, sNumSeq, (int64_t)ra.sLo) != seq_of(sSeqStart, sNumSeq, (int64_t)rb.sLo))
        return;
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

// a chain's extent, accumulated on its root: lowest source position, total length (the lowest target position is the root's own)
static __global__ void __launch_bounds__(256) k_merge_extent(const ComposedRec<int32_t> *__restrict__ recs, const uint32_t *__restrict__ root, uint32_t n,
                                                             uint32_t *__restrict__ minS, uint32_t *__restrict__ sumLen) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t r = root[i];
    atomicMin(&minS[r], (uint32_t)recs[i].sLo);
    atomicAdd(&sumLen[r], (uint32_t)recs[i].len);
}

// sort keys of the chains: (source start, target start); pieces that are not a chain's root sort behind everything
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

// merged records in (source start, target start) order.  mEncF = target strand | target sequence << 8
static __global__ void __launch_bounds__(256) k_merge_records(const ComposedRec<int32_t> *__restrict__ recs, const uint32_t *__restrict__ sortedHeads,
                                                              uint32_t m, const uint32_t *__restrict__ minS, const uint32_t *__restrict__ sumLen,
                                                              const int64_t *__restrict__ tSeqStart, int tNumSeq,
                                                              ComposedRec<int32_t> *__restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m)
        return;
    const uint32_t h = sortedHeads[j];
    const ComposedRec<int32_t> r = recs[h];
    ComposedRec<int32_t> o;
    o.sLo = (int32_t)minS[h];
    o.len = (int32_t)sumLen[h];
    o.so = r.so;
    const uint32_t seq = tNumSeq > 1 ? (uint32_t)seq_of(tSeqStart, tNumSeq, (int64_t)r.so) : 0u;
    o.mEncF = (r.mEncF & 1u) | (seq << 8);
    out[j] = o;
}

// ---- flags: target overlap with a record whose source lies within `window` bases ----
static __global__ void __launch_bounds__(256) k_flag_keys(const ComposedRec<int32_t> *__restrict__ recs, uint32_t m, uint64_t *__restrict__ keys) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m)
        keys[j] = ((uint64_t)(uint32_t)recs[j].so << 32) | (uint64_t)j; // target start, record index
}
// high target ends in target-start order (input of the running maximum)
static __global__ void __launch_bounds__(256) k_flag_ends(const ComposedRec<int32_t> *__restrict__ recs, const uint64_t *__restrict__ sortedKeys, uint32_t m,
                                                          uint32_t *__restrict__ tHiSorted) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m)
        return;
    const ComposedRec<int32_t> x = recs[(uint32_t)sortedKeys[r]];
    tHiSorted[r] = (uint32_t)x.so + (uint32_t)x.len - 1u;
}
struct MaxOp {
    __host__ __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};
static constexpr int FLAG_SCAN_CUT = 128; // neighbours looked at in either direction before giving up (and flagging)
// Every record looks at the records after it in target-start order while they begin inside its target range, and at the
// records before it while any of them can still reach it (running maximum of the high ends).  A pair is seen from its
// earlier member unless that scan was cut off — then the earlier member is flagged as it stands and the later member's
// backward scan either finds the pair or is cut off and flags itself.
static __global__ void __launch_bounds__(256) k_flag_overlaps(const ComposedRec<int32_t> *__restrict__ recs, const uint64_t *__restrict__ sortedKeys,
                                                              const uint32_t *__restrict__ runMax, uint32_t m, int64_t window,
                                                              uint32_t *__restrict__ flag) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m)
        return;
    const uint32_t i = (uint32_t)sortedKeys[r];
    const ComposedRec<int32_t> x = recs[i];
    const int64_t tLo = (int64_t)(uint32_t)x.so, tHi = tLo + x.len - 1, sLo = x.sLo, sHi = sLo + x.len - 1;
    auto near = [&](const ComposedRec<int32_t> &y) {
        const int64_t yLo = y.sLo, yHi = yLo + y.len - 1;
        const int64_t gap = (sLo > yLo ? sLo : yLo) - (sHi < yHi ? sHi : yHi);
        return gap < window;
    };
    bool mine = false;
    int steps = 0;
    for (uint32_t r2 = r + 1; r2 < m; ++r2) {
        const uint32_t j = (uint32_t)sortedKeys[r2];
        if ((int64_t)(sortedKeys[r2] >> 32) > tHi)
            break;
        if (++steps > FLAG_SCAN_CUT) {
            mine = true;
            break;
        }
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
        const uint32_t j = (uint32_t)sortedKeys[r2];
        const ComposedRec<int32_t> y = recs[j];
        if ((int64_t)(uint32_t)y.so + y.len - 1 >= tLo && near(y)) {
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
static __global__ void __launch_bounds__(256) k_merge_mark(ComposedRec<int32_t> *__restrict__ recs, const uint32_t *__restrict__ flag, uint32_t m) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) {
        if (flag[j])
            recs[j].mEncF |= 2u;
    } else if (j < m + LIFT_SENTINELS) {
        ComposedRec<int32_t> o;
        o.sLo = 0x7FFFFFFF;
        o.len = 0;
        o.so = 0;
        o.mEncF = 0;
        recs[j] = o;
    }
}

} // namespace hgx
