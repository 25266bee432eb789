// Kernels that build a composed table on the device (buildComposed, hgx_liftover.hip): the walk kernels have lifted every
// source top segment as one interval and left the pieces in a frontier; they are sorted by source position (radix sort on
// the position, the piece's frontier slot as payload) and turned into ComposedRecs and the two bucket tables.
#pragma once
#include "hgx_liftover_kernels.hpp"

namespace hgx {

// one whole-segment interval per source top segment, forward strand
template <typename C>
static __global__ void __launch_bounds__(256) k_table_queries(const TopRec<C> *__restrict__ top, uint32_t nt, int64_t *__restrict__ gs,
                                                              int64_t *__restrict__ ge, uint8_t *__restrict__ st) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) {
        gs[i] = (int64_t)top[i].start;
        ge[i] = (int64_t)top[i + 1].start - 1;
        st[i] = (uint8_t)'+';
    }
}

// dense view of the captured frontier: sort key (source position) and physical slot of every piece
static __global__ void __launch_bounds__(256) k_table_keys(Frontier in, const unsigned long long *inCount, uint32_t cap,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ slots) {
    __shared__ FrontView fview;
    const uint32_t n = front_view_init(&fview, inCount, cap);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t p = front_slot(&fview, i, cap);
        keys[i] = (uint64_t)in.sPos[p];
        slots[i] = p;
    }
}

// records in source order.  through: the pieces are FINAL (so = forward target start); otherwise they are bottom pieces in
// the ancestor (index + offset) and eo[] receives the bases of the bottom segment after the piece.
template <typename C>
static __global__ void __launch_bounds__(256) k_table_records(Frontier in, const uint32_t *__restrict__ sortedSlots, uint32_t n, int through,
                                                              const BotRec<C> *__restrict__ mbot, ComposedRec<C> *__restrict__ recs,
                                                              C *__restrict__ eo, unsigned long long *bad) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const uint32_t p = sortedSlots[k];
    const uint8_t fl = in.flags[p];
    if (fl & F_SREV)
        *bad = 1; // a forward source segment never yields a source-reversed piece
    const int32_t qid = in.qid[p];
    const int32_t prevQ = k ? in.qid[sortedSlots[k - 1]] : -1; // pieces of one segment are adjacent (segments are disjoint)
    ComposedRec<C> r{};
    r.sLo = (C)in.sPos[p];
    r.len = (C)in.len[p];
    r.so = (C)in.so[p];
    uint32_t enc = (qid != prevQ ? 2u : 0u) | ((fl & F_TREV) ? 1u : 0u);
    if (!through) {
        const int32_t idx = in.idx[p];
        enc |= (uint32_t)idx << 2;
        eo[k] = (C)(((int64_t)mbot[idx + 1].start - (int64_t)mbot[idx].start) - (int64_t)r.so - (int64_t)r.len);
    }
    r.mEncF = enc;
    recs[k] = r;
}

// starts[b] = first record that begins at or after position b << shift, b = 0 .. nb
template <typename C>
static __global__ void __launch_bounds__(256) k_table_starts(const ComposedRec<C> *__restrict__ recs, uint32_t n, int shift, uint32_t nb,
                                                             uint32_t *__restrict__ starts) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb)
        return;
    const int64_t pos = (int64_t)b << shift;
    uint32_t lo = 0, hi = n; // first k with recs[k].sLo >= pos
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)recs[mid].sLo < pos)
            lo = mid + 1;
        else
            hi = mid;
    }
    starts[b] = lo;
}

// coarse[b] = the first record that touches bucket b (records come in source order, so it is the smallest index) ...
template <typename C>
static __global__ void __launch_bounds__(256) k_table_touch(const ComposedRec<C> *__restrict__ recs, uint32_t n, int shift,
                                                            uint32_t *__restrict__ coarse) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    const int64_t lo = (int64_t)recs[k].sLo, hi = lo + (int64_t)recs[k].len - 1;
    for (int64_t b = lo >> shift; b <= (hi >> shift); ++b)
        atomicMin(&coarse[b], k);
}
// ... or, where no record touches the bucket, the first one that begins after it
static __global__ void __launch_bounds__(256) k_table_fill(uint32_t *__restrict__ coarse, const uint32_t *__restrict__ starts, uint32_t nb) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nb && coarse[b] == 0xFFFFFFFFu)
        coarse[b] = starts[b];
}

// ---- per-genome side tables of the walk kernels, derived on the device from the uploaded TopRec / BotRec tables (a plan's
// creation used to spend tens of milliseconds building them in host loops) ----
// ChainRec of every top segment (hgx_device.hpp): `last` selects the parent bottom segment's own index as the link instead
// of its top-parse index
template <typename C>
static __global__ void __launch_bounds__(256) k_make_chain(const TopRec<C> *__restrict__ top, const BotRec<C> *__restrict__ pbot, uint32_t numTop,
                                                           int last, ChainRec<C> *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numTop)
        return;
    const TopRec<C> t = top[i];
    const int64_t len = (int64_t)top[i + 1].start - (int64_t)t.start;
    ChainRec<C> r;
    if (t.parentEnc < 0) {
        r.set((int64_t)t.start, 0, len, false, 0, false);
    } else {
        const int32_t p = t.parentEnc >> 1;
        const BotRec<C> b = pbot[p];
        const int64_t link = last ? (int64_t)p : (int64_t)(b.topParse < 0 ? 0 : b.topParse);
        r.set((int64_t)t.start, (int64_t)b.start, len, true, link, (t.parentEnc & 1) != 0);
    }
    out[i] = r;
}
// DownRec of every bottom segment for one child slot
template <typename C>
static __global__ void __launch_bounds__(256) k_make_down(const int32_t *__restrict__ childEnc, const TopRec<C> *__restrict__ ctop, uint32_t numBot,
                                                          DownRec<C> *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numBot)
        return;
    DownRec<C> r;
    memset(&r, 0, sizeof r);
    r.childEnc = childEnc[i];
    r.paralogy = -1;
    if (r.childEnc >= 0) {
        const int32_t c = r.childEnc >> 1;
        const TopRec<C> t = ctop[c];
        r.childStart = t.start;
        r.len = (C)((int64_t)ctop[c + 1].start - (int64_t)t.start);
        r.paralogy = t.paralogy;
    }
    out[i] = r;
}
// coarse position -> segment table: out[b] = index of the segment that holds position b << shift (b < nb), out[nb] = last segment
template <typename REC>
static __global__ void __launch_bounds__(256) k_make_locate(const REC *__restrict__ segs, int64_t nseg, int shift, uint32_t nb,
                                                            int32_t *__restrict__ out) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb)
        return;
    if (b == nb) {
        out[b] = (int32_t)(nseg - 1);
        return;
    }
    const int64_t pos = (int64_t)b << shift;
    int64_t lo = 0, hi = nseg; // last segment with start <= pos
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)segs[mid].start <= pos)
            lo = mid;
        else
            hi = mid;
    }
    out[b] = (int32_t)lo;
}

} // namespace hgx
