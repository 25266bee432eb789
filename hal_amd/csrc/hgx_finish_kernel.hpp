// Per-interval finishing kernel: turns the unordered bag of mapped pieces of one query interval into
// the reference's output lines.  One wavefront per interval, data staged in LDS.
//
//  1. order pieces like the reference's MappedSegmentSet: by target, then source, in forward
//     coordinates (MappedSegment::lessThan / fastComp, api/impl/halMappedSegment.cpp:36-43,167-206);
//  2. break overlaps: insertAndBreakOverlaps (api/impl/halSegmentMapper.cpp:475-520) leaves a set in which
//     any two members have identical or disjoint target ranges, i.e. every piece cut at every boundary of
//     every piece overlapping it — computed here directly as that common refinement (order independent),
//     source side sliced in lock-step (MappedSegment::slice, halMappedSegment.cpp:395-402);
//  3. merge runs: BlockMapper::extractSegment (liftover/impl/halBlockMapper.cpp:331-394) with
//     canMergeRightWith (halMappedSegment.cpp:109-161) and the query cut-point set, walking the set in
//     target order exactly as liftInterval does (liftover/impl/halBlockLiftover.cpp:79-105);
//  4. stable sort of the lines by source start (Liftover::visitLine, liftover/impl/halLiftover.cpp:90).
// Queries whose working set exceeds the staging capacity are deferred to the same code running on a
// global-memory scratch area (template parameter); nothing is approximated.
#pragma once
#include "../../include/hgx.h"
#include "hgx_liftover_kernels.hpp"

namespace hgx {

// -DHGX_LIFT_PROFILE=4 (make profile-lib PROFILE_KERNEL=4): the cycles finish_query spends in each of its phases, summed over all
// intervals of the finishing kernels (lane 0 books; hgx_liftover.hip prints the sums after a single-pass run)
#if defined(HGX_LIFT_PROFILE) && HGX_LIFT_PROFILE == 4
__device__ unsigned long long g_finishProfile[16];
#define FQ_PROF_DECL unsigned long long fqT = __builtin_readcyclecounter()
#define FQ_PROF(i)                                                                                                                           \
    {                                                                                                                                        \
        const unsigned long long now = __builtin_readcyclecounter();                                                                         \
        if (lane_id() == 0)                                                                                                                  \
            atomicAdd(&g_finishProfile[i], now - fqT);                                                                                       \
        fqT = now;                                                                                                                           \
    }
#else
#define FQ_PROF_DECL
#define FQ_PROF(i)
#endif

template <typename C> struct FinishStore {
    // piece arrays, two banks (A = current, B = scratch / refinement output)
    C *tLo, *tHi, *sLo, *sHi;
    uint8_t *fl;
    C *tLo2, *tHi2, *sLo2, *sHi2;
    uint8_t *fl2;
    int32_t *seq;   // target sequence index of each piece
    uint32_t *ord;  // permutation / offsets scratch, capacity cap2 (power of two >= 2*cap)
    C *bnd, *bnd2;  // boundary coordinates (raw / sorted unique), capacity cap2 each
    uint8_t *alive; // set membership during extraction
    C *cut;         // query cut points
    // output lines
    C *lStart, *lEnd, *lSrc;
    int32_t *lSeq;
    uint8_t *lStrand;
    int cap, cap2, cutCap;
    int blocks = 0; // 1: return the refined set itself (BlockMapper::getMap), no extractSegment merging; 2: the same without
                    // the refinement either (the mapped pieces, sorted, equal ones once)
};

__device__ __forceinline__ void wsync() {
    __syncthreads(); // block == one wavefront
}

// bitonic sort of ord[0..n2) (n2 power of two) by a strict weak order on the indices
template <typename Less> __device__ __forceinline__ void wave_bitonic(uint32_t *ord, int n2, Less less) {
    const int lane = lane_id();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i + j;
                const bool up = (i & k) == 0;
                const uint32_t a = ord[i], b = ord[p];
                const bool swap = up ? less(b, a) : less(a, b);
                if (swap) {
                    ord[i] = b;
                    ord[p] = a;
                }
            }
            wsync();
        }
    }
}

static constexpr int RANK_SORT_MAX = 128; // sets up to this size are sorted by counting ranks (quadratic: beyond it the bitonic network)

__device__ __forceinline__ int pow2_at_least(int n) {
    int p = 1;
    while (p < n)
        p <<= 1;
    return p;
}

// the two sorts by ranks counted in registers (defined behind the lane-exchange helpers, below)
template <typename C, int REGS> __device__ __forceinline__ void rank_sort_regs(FinishStore<C> &S, int n);
template <typename C, int REGS> __device__ __forceinline__ void rank_lines_regs(FinishStore<C> &S, int nl);

// sort bank A by (tLo, tHi, sLo, sHi) into bank B, then swap the banks
template <typename C, int REGS = 0> __device__ __forceinline__ void sort_pieces(FinishStore<C> &S, int n) {
    const int lane = lane_id();
    if constexpr (REGS > 0) {
        if (n <= 64 * REGS && n <= RANK_SORT_MAX) { // (wave-uniform)
            rank_sort_regs<C, REGS>(S, n);
            return;
        }
    }
    if (n <= RANK_SORT_MAX) {
        // a rank by counting: every lane compares its piece with piece j, read by all lanes at once (an LDS broadcast), j = 0 .. n - 1,
        // and writes it at its rank (equal keys: by index) — for a hundred pieces a tenth of the bitonic network's LDS traffic and
        // none of its 36 barriers (the sorts were 60 % of an interval's time once extractSegment was off the serial loop)
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            const bool in = i < n;
            const C mt = in ? S.tLo[i] : (C)0, mh = in ? S.tHi[i] : (C)0, ms = in ? S.sLo[i] : (C)0, me = in ? S.sHi[i] : (C)0;
            const uint8_t mf = in ? S.fl[i] : (uint8_t)0;
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const C a = S.tLo[j], b = S.tHi[j], c = S.sLo[j], d = S.sHi[j];
                const bool less = a != mt ? a < mt : b != mh ? b < mh : c != ms ? c < ms : d != me ? d < me : j < i;
                rank += less ? 1 : 0;
            }
            if (in) {
                S.tLo2[rank] = mt;
                S.tHi2[rank] = mh;
                S.sLo2[rank] = ms;
                S.sHi2[rank] = me;
                S.fl2[rank] = mf;
            }
        }
        wsync();
        C *t;
        uint8_t *u;
        t = S.tLo, S.tLo = S.tLo2, S.tLo2 = t;
        t = S.tHi, S.tHi = S.tHi2, S.tHi2 = t;
        t = S.sLo, S.sLo = S.sLo2, S.sLo2 = t;
        t = S.sHi, S.sHi = S.sHi2, S.sHi2 = t;
        u = S.fl, S.fl = S.fl2, S.fl2 = u;
        return;
    }
    const int n2 = pow2_at_least(n);
    for (int i = lane; i < n2; i += 64)
        S.ord[i] = (uint32_t)i;
    wsync();
    const C *tLo = S.tLo, *tHi = S.tHi, *sLo = S.sLo, *sHi = S.sHi;
    wave_bitonic(S.ord, n2, [=](uint32_t a, uint32_t b) {
        if ((int)a >= n)
            return false;
        if ((int)b >= n)
            return true;
        if (tLo[a] != tLo[b])
            return tLo[a] < tLo[b];
        if (tHi[a] != tHi[b])
            return tHi[a] < tHi[b];
        if (sLo[a] != sLo[b])
            return sLo[a] < sLo[b];
        return sHi[a] < sHi[b];
    });
    for (int i = lane; i < n; i += 64) {
        const uint32_t o = S.ord[i];
        S.tLo2[i] = S.tLo[o];
        S.tHi2[i] = S.tHi[o];
        S.sLo2[i] = S.sLo[o];
        S.sHi2[i] = S.sHi[o];
        S.fl2[i] = S.fl[o];
    }
    wsync();
    C *t;
    uint8_t *u;
    t = S.tLo, S.tLo = S.tLo2, S.tLo2 = t;
    t = S.tHi, S.tHi = S.tHi2, S.tHi2 = t;
    t = S.sLo, S.sLo = S.sLo2, S.sLo2 = t;
    t = S.sHi, S.sHi = S.sHi2, S.sHi2 = t;
    u = S.fl, S.fl = S.fl2, S.fl2 = u;
}

// extractSegment with the set spread over the lanes' registers (defined behind the lane-exchange helpers, below)
template <typename C, int REGS> __device__ __forceinline__ void extract_segment_regs(FinishStore<C> &S, int n, int &nlOut, int &failOut);

// Returns the number of output lines written to S.l* (ordered for printing through S.ord), or -need
// (need > 0) when the staging capacity is insufficient; `need` is a lower bound of the capacity to retry with.
// REGS > 0: a set of up to 64 * REGS members goes through extractSegment in registers (extract_segment_regs) instead of the
// serial loop over the staging arrays.
template <typename C, int REGS = 0>
__device__ __forceinline__ int finish_query(FinishStore<C> &S, const Mapped &in, uint32_t base, int n, const int64_t *__restrict__ seqStart,
                                            int numSeq) {
    const int lane = lane_id();
    FQ_PROF_DECL;
    if (n > S.cap)
        return -n;
    for (int i = lane; i < n; i += 64) {
        const MappedRec r = in.rec[base + i];
        S.tLo[i] = (C)r.tLo;
        S.tHi[i] = (C)(r.tLo + r.len - 1);
        S.sLo[i] = (C)r.sLo;
        S.sHi[i] = (C)(r.sLo + r.len - 1);
        S.fl[i] = (uint8_t)r.flags;
    }
    wsync();
    FQ_PROF(8) // load
    sort_pieces<C, REGS>(S, n);
    FQ_PROF(0) // first sort

    // does any pair of neighbours (in target order) overlap without having the same target range?
    // (if any two pieces do, two neighbours do); are there exact duplicates?
    bool refine = false, dup = false;
    for (int i = lane; i + 1 < n; i += 64) {
        const bool same = S.tLo[i + 1] == S.tLo[i] && S.tHi[i + 1] == S.tHi[i];
        if (S.tLo[i + 1] <= S.tHi[i] && !same)
            refine = true;
        if (same && S.sLo[i + 1] == S.sLo[i] && S.sHi[i + 1] == S.sHi[i])
            dup = true;
    }
    refine = __any(refine) && S.blocks != 2; // (blocks == 2: the pieces as halMapSegment's walk leaves them, before insertAndBreakOverlaps)
    dup = __any(dup);

    if (refine) {
        // cut coordinates: every piece start and every piece end + 1
        if (2 * n > S.cap2)
            return -(2 * n);
        const int nb2 = pow2_at_least(2 * n);
        for (int i = lane; i < n; i += 64) {
            S.bnd[2 * i] = S.tLo[i];
            S.bnd[2 * i + 1] = S.tHi[i] + 1;
        }
        for (int i = lane; i < nb2; i += 64)
            S.ord[i] = (uint32_t)i;
        wsync();
        {
            const C *bnd = S.bnd;
            const int nn = 2 * n;
            wave_bitonic(S.ord, nb2, [=](uint32_t a, uint32_t b) {
                if ((int)a >= nn)
                    return false;
                if ((int)b >= nn)
                    return true;
                return bnd[a] < bnd[b];
            });
        }
        FQ_PROF(1) // boundaries sorted
        // sorted copy, then the distinct values, 64 at a time: a value that differs from the one in front of it is kept, its
        // place is the number of such values before it (ballot + popcount, a running base) — into the raw array, which is free
        for (int i = lane; i < 2 * n; i += 64)
            S.bnd2[i] = S.bnd[S.ord[i]];
        wsync();
        int nbu = 0;
        for (int c0 = 0; c0 < 2 * n; c0 += 64) {
            const int i = c0 + lane;
            const bool in2 = i < 2 * n;
            const C v = in2 ? S.bnd2[i] : (C)0;
            const bool fresh = in2 && (i == 0 || v != S.bnd2[i - 1]);
            const unsigned long long m = __ballot(fresh);
            if (fresh)
                S.bnd[nbu + (int)__popcll(m & ((1ull << lane) - 1ull))] = v;
            nbu += (int)__popcll(m);
        }
        wsync();
        {
            C *t = S.bnd;
            S.bnd = S.bnd2;
            S.bnd2 = t;
        }
        // number of sub-pieces of piece i = 1 + #{x in bnd : tLo < x <= tHi}
        auto upper = [&](C v) { // first index with bnd[idx] > v
            int lo = 0, hi = nbu;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (S.bnd2[mid] <= v)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            return lo;
        };
        for (int i = lane; i < n; i += 64)
            S.ord[i] = (uint32_t)(1 + upper(S.tHi[i]) - upper(S.tLo[i]));
        wsync();
        int m = 0; // exclusive prefix sums, 64 at a time with a running base
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            const uint32_t c = i < n ? S.ord[i] : 0u;
            uint32_t incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if (lane >= o)
                    incl += up;
            }
            if (i < n)
                S.ord[i] = (uint32_t)m + incl - c;
            m += __builtin_amdgcn_readlane((int)incl, 63);
        }
        wsync();
        if (m > S.cap)
            return -m;
        FQ_PROF(2) // distinct boundaries, counts, prefix sums
        // emit the refined pieces into bank B
        for (int i = lane; i < n; i += 64) {
            int o = (int)S.ord[i];
            const C tl = S.tLo[i], th = S.tHi[i], sl = S.sLo[i], sh = S.sHi[i];
            const uint8_t f = S.fl[i];
            const bool opposite = ((f & F_SREV) != 0) != ((f & F_TREV) != 0);
            int k = upper(tl);
            C lo = tl;
            for (;;) {
                const C hi = (k < nbu && S.bnd2[k] <= th) ? (C)(S.bnd2[k] - 1) : th;
                S.tLo2[o] = lo;
                S.tHi2[o] = hi;
                if (!opposite) {
                    S.sLo2[o] = sl + (lo - tl);
                    S.sHi2[o] = sl + (hi - tl);
                } else {
                    S.sLo2[o] = sh - (hi - tl);
                    S.sHi2[o] = sh - (lo - tl);
                }
                S.fl2[o] = f;
                ++o;
                if (hi == th)
                    break;
                lo = hi + 1;
                ++k;
            }
        }
        wsync();
        {
            C *t;
            uint8_t *u;
            t = S.tLo, S.tLo = S.tLo2, S.tLo2 = t;
            t = S.tHi, S.tHi = S.tHi2, S.tHi2 = t;
            t = S.sLo, S.sLo = S.sLo2, S.sLo2 = t;
            t = S.sHi, S.sHi = S.sHi2, S.sHi2 = t;
            u = S.fl, S.fl = S.fl2, S.fl2 = u;
        }
        n = m;
        FQ_PROF(3) // refined pieces emitted
        sort_pieces<C, REGS>(S, n);
        FQ_PROF(4) // second sort
        dup = false;
        for (int i = lane; i + 1 < n; i += 64)
            if (S.tLo[i + 1] == S.tLo[i] && S.tHi[i + 1] == S.tHi[i] && S.sLo[i + 1] == S.sLo[i] && S.sHi[i + 1] == S.sHi[i])
                dup = true;
        dup = __any(dup);
    }
    if (dup) { // std::set keeps the first of equal keys: a member equal to the one in front of it goes (equal keys are neighbours)
        int w = 0;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            const bool keep = i < n && (i == 0 || S.tLo[i] != S.tLo[i - 1] || S.tHi[i] != S.tHi[i - 1] || S.sLo[i] != S.sLo[i - 1] ||
                                        S.sHi[i] != S.sHi[i - 1]);
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const int o = w + (int)__popcll(km & ((1ull << lane) - 1ull));
                S.tLo2[o] = S.tLo[i];
                S.tHi2[o] = S.tHi[i];
                S.sLo2[o] = S.sLo[i];
                S.sHi2[o] = S.sHi[i];
                S.fl2[o] = S.fl[i];
            }
            w += (int)__popcll(km);
        }
        wsync();
        {
            C *t;
            uint8_t *u;
            t = S.tLo, S.tLo = S.tLo2, S.tLo2 = t;
            t = S.tHi, S.tHi = S.tHi2, S.tHi2 = t;
            t = S.sLo, S.sLo = S.sLo2, S.sLo2 = t;
            t = S.sHi, S.sHi = S.sHi2, S.sHi2 = t;
            u = S.fl, S.fl = S.fl2, S.fl2 = u;
        }
        n = w;
    }

    // target sequence of each piece (Segment::getSequence: site -> sequence, binary search on start[])
    for (int i = lane; i < n; i += 64) {
        int s = 0;
        if (numSeq > 1) {
            int lo = 0, hi = numSeq;
            const int64_t pos = (int64_t)S.tLo[i];
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (seqStart[mid] <= pos)
                    lo = mid;
                else
                    hi = mid;
            }
            s = lo;
        }
        S.seq[i] = s;
        S.alive[i] = 1;
    }
    wsync();

    if (S.blocks) {
        // BlockMapper::getMap (liftover/inc/halBlockMapper.h:36): the set after insertAndBreakOverlaps, in set order
        // (target, then source).  One line per member: forward target range, forward source start, both orientations.
        for (int i = lane; i < n; i += 64) {
            S.lStart[i] = S.tLo[i];
            S.lEnd[i] = S.tHi[i] + 1;
            S.lSrc[i] = S.sLo[i];
            S.lSeq[i] = S.seq[i];
            S.lStrand[i] = (uint8_t)(((S.fl[i] & F_SREV) ? '-' : '+') | ((S.fl[i] & F_TREV) ? 0x80 : 0));
            S.ord[i] = (uint32_t)i;
        }
        wsync();
        return n;
    }

    FQ_PROF(5) // duplicates, sequences
    // ---- extractSegment over the set in target order (in registers when the set fits them, else serial: lane 0) ----
    int nl = 0;
    int fail = 0;
    bool inRegs = false;
    if constexpr (REGS > 0) {
        if (n <= 64 * REGS) { // (wave-uniform)
            extract_segment_regs<C, REGS>(S, n, nl, fail);
            inRegs = fail != 2;
            if (!inRegs)
                nl = fail = 0;
        }
    }
    if (!inRegs && lane == 0) {
        int ncut = 0;
        auto nextAlive = [&](int i) {
            while (i < n && !S.alive[i])
                ++i;
            return i;
        };
        auto canMergeRight = [&](int a, int b) { // halMappedSegment.cpp:109-161 in forward coordinates
            const uint8_t fa = S.fl[a], fb = S.fl[b];
            if (((fa ^ fb) & (F_SREV | F_TREV)) != 0)
                return false;
            if (S.tLo[b] - S.tHi[a] != 1)
                return false;
            const bool same = ((fa & F_SREV) != 0) == ((fa & F_TREV) != 0);
            const bool rOk = same ? (S.sLo[b] - S.sHi[a] == 1) : (S.sLo[a] - S.sHi[b] == 1);
            if (!rOk)
                return false;
            const C cutPos = S.tHi[a];
            for (int c = 0; c < ncut; ++c)
                if (S.cut[c] == cutPos)
                    return false;
            return true;
        };
        for (int i = 0; i < n && !fail; ++i) {
            if (!S.alive[i])
                continue;
            int v1s = i, v1n = 1, back = i;
            int nxt = nextAlive(i + 1);
            while (nxt < n && S.tLo[back] == S.tLo[nxt]) {
                back = nxt;
                ++v1n;
                nxt = nextAlive(nxt + 1);
            }
            int fragBack = i;
            while (nxt < n) {
                const int v2s = nxt;
                int v2n = 0, b2 = -1;
                while (nxt < n && (v2n == 0 || S.tLo[b2] == S.tLo[nxt]) && v2n < v1n) {
                    b2 = nxt;
                    ++v2n;
                    nxt = nextAlive(nxt + 1);
                }
                bool can = v1n == v2n;
                int a = v1s, b = v2s;
                for (int k = 0; k < v1n && can; ++k) {
                    can = S.seq[b] == S.seq[i] && canMergeRight(a, b);
                    a = nextAlive(a + 1);
                    b = nextAlive(b + 1);
                }
                if (!can)
                    break;
                fragBack = v2s;
                S.alive[v2s] = 0; // erased from the set (halBlockMapper.cpp:389-391)
                v1s = v2s;
                v1n = v2n;
            }
            if (v1n > 1) {
                if (ncut == S.cutCap) {
                    fail = 1;
                    break;
                }
                S.cut[ncut++] = S.tHi[fragBack];
            }
            // halBlockLiftover.cpp:82-105
            const C a0 = S.tLo[i] < S.tLo[fragBack] ? S.tLo[i] : S.tLo[fragBack];
            const C a1 = S.tHi[i] > S.tHi[fragBack] ? S.tHi[i] : S.tHi[fragBack];
            S.lStart[nl] = a0;
            S.lEnd[nl] = a1 + 1;
            S.lSrc[nl] = S.sLo[i] < S.sLo[fragBack] ? S.sLo[i] : S.sLo[fragBack];
            S.lSeq[nl] = S.seq[i];
            // strand character in the low 7 bits, the piece's own orientation in bit 7
            S.lStrand[nl] = (uint8_t)(((S.fl[i] & F_DOT) ? '.' : ((S.fl[i] & F_TREV) ? '-' : '+')) | ((S.fl[i] & F_TREV) ? 0x80 : 0));
            ++nl;
        }
        S.ord[0] = (uint32_t)nl;
        S.ord[1] = (uint32_t)fail;
    }
    wsync();
    if (!inRegs) {
        nl = (int)S.ord[0];
        fail = (int)S.ord[1];
    }
    wsync();
    if (fail)
        return -(4 * S.cap);
    FQ_PROF(6) // extractSegment
    // stable sort by source start: key (lSrc, line index)
    if constexpr (REGS > 0) {
        if (nl <= 64 * REGS && nl <= RANK_SORT_MAX) {
            rank_lines_regs<C, REGS>(S, nl);
            FQ_PROF(7) // lines sorted
            return nl;
        }
    }
    if (nl <= RANK_SORT_MAX) { // (ranks by counting, as in sort_pieces: ord[rank] = line)
        for (int c0 = 0; c0 < nl; c0 += 64) {
            const int i = c0 + lane;
            const C mine = i < nl ? S.lSrc[i] : (C)0;
            int rank = 0;
            for (int j = 0; j < nl; ++j) {
                const C a = S.lSrc[j];
                rank += (a != mine ? a < mine : j < i) ? 1 : 0;
            }
            if (i < nl)
                S.ord[rank] = (uint32_t)i;
        }
        wsync();
        FQ_PROF(7) // lines sorted
        return nl;
    }
    const int l2 = pow2_at_least(nl);
    for (int i = lane; i < l2; i += 64)
        S.ord[i] = (uint32_t)i;
    wsync();
    {
        const C *lSrc = S.lSrc;
        const int nn = nl;
        wave_bitonic(S.ord, l2, [=](uint32_t a, uint32_t b) {
            if ((int)a >= nn)
                return false;
            if ((int)b >= nn)
                return true;
            if (lSrc[a] != lSrc[b])
                return lSrc[a] < lSrc[b];
            return a < b;
        });
    }
    FQ_PROF(7) // lines sorted
    return nl;
}

template <typename C>
__device__ __forceinline__ void write_records(const FinishStore<C> &S, int nl, int32_t q, hgx_record *__restrict__ dst,
                                              const int64_t *__restrict__ seqStart) {
    for (int k = lane_id(); k < nl; k += 64) {
        const uint32_t o = S.ord[k];
        hgx_record r;
        const int32_t s = S.lSeq[o];
        const int64_t ss = seqStart[s];
        r.query = q;
        r.tgt_start = (int64_t)S.lStart[o] - ss;
        r.tgt_end = (int64_t)S.lEnd[o] - ss;
        r.src_start = (int64_t)S.lSrc[o];
        r.tgt_seq = s;
        r.strand = (char)(S.lStrand[o] & 0x7F);
        r.tgt_reversed = (uint8_t)(S.lStrand[o] >> 7);
        r._pad[0] = r._pad[1] = 0;
        dst[k] = r;
    }
}


// ---------------------------------------------------------------------------------------------
// Size classes.  Most intervals map to a handful of pieces; those are finished in registers by a sub-wave
// group of G lanes (k_finish_fast<G>), G the smallest of 8/16/32/64 that holds them: each instantiation scans
// all intervals and handles the ones of its class.  Larger intervals, and any interval whose pieces overlap or
// tie on the target (the cases that need overlap breaking / equivalence classes), are appended to `generalList`
// for the general LDS kernel.
// Lane exchanges of the register-resident kernels.  ds_bpermute (what __shfl* compiles to) goes through the LDS crossbar;
// the fixed patterns of a sorting network inside a quad or a row of 16 lanes, and the shift by one lane, are DPP modifiers
// of an ordinary VALU move on gfx9-family parts: xor 1 / 2 / 3 = quad_perm, xor 7 = row_half_mirror, xor 15 = row_mirror,
// previous lane = wave_shr:1.  k_finish_fast<C, 8> went from 53 ds_bpermute per iteration to 9 (with the narrower gathers
// below) — and kept its 0.10 ms per 1 M intervals: the kernel is bound by the number of instructions its eight wavefronts
// per SIMD issue (VALU 47 % busy plus scalar and LDS work, profiles/r01q_pmc.txt), not by the crossbar.  (32-bit sort
// keys — position << log2 G | lane when both fit — cut the static instruction count of the G = 16..64 instantiations by
// 30-40 % and changed nothing either; not kept.  Prefetching the next iteration's count and offset: nothing.  What the
// wavefronts wait for 70 % of their life is not settled; the 40-byte records leave as five 8-byte stores per lane.)
template <int CTRL> __device__ __forceinline__ int dpp_move(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); // lanes without a source keep their own value
}
template <int CTRL> __device__ __forceinline__ unsigned long long dpp_move(unsigned long long v) {
    const int lo = dpp_move<CTRL>((int)(uint32_t)v), hi = dpp_move<CTRL>((int)(uint32_t)(v >> 32));
    return ((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo;
}
template <int CTRL> __device__ __forceinline__ int64_t dpp_move(int64_t v) {
    return (int64_t)dpp_move<CTRL>((unsigned long long)v);
}
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_XOR3 = 0x1B, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140, DPP_WAVE_SHR1 = 0x138 };
// value of lane ^ M
template <int M> __device__ __forceinline__ unsigned long long lane_xor(unsigned long long v) {
    if (M == 1)
        return dpp_move<DPP_XOR1>(v);
    if (M == 2)
        return dpp_move<DPP_XOR2>(v);
    if (M == 3)
        return dpp_move<DPP_XOR3>(v);
    if (M == 7)
        return dpp_move<DPP_HALF_MIRROR>(v);
    if (M == 15)
        return dpp_move<DPP_ROW_MIRROR>(v);
    return __shfl_xor(v, M);
}
// value of the previous lane (lane 0: its own)
template <typename T> __device__ __forceinline__ T lane_prev(T v) {
    return dpp_move<DPP_WAVE_SHR1>(v);
}

// a position (below 2^31 when C is int32) from another lane
template <typename C> __device__ __forceinline__ int64_t shfl_pos(int64_t v, int srcLane) {
    if (sizeof(C) == 4)
        return (int64_t)__shfl((int)v, srcLane);
    return __shfl(v, srcLane);
}

// one merge level of the sorting network: blocks of K lanes, each made of two ascending halves, become ascending.  The
// first step compares lane i with lane i ^ (K - 1) (the mirror image inside the block), the others i with i ^ j.
template <int K, int J> struct MergeSteps {
    static __device__ __forceinline__ unsigned long long run(unsigned long long key, int li) {
        const unsigned long long other = lane_xor<J>(key);
        const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
        key = (li & J) == 0 ? mn : mx;
        return MergeSteps<K, (J >> 1)>::run(key, li);
    }
};
template <int K> struct MergeSteps<K, 0> {
    static __device__ __forceinline__ unsigned long long run(unsigned long long key, int) { return key; }
};
template <int K> __device__ __forceinline__ unsigned long long merge_level(unsigned long long key, int li) {
    const unsigned long long other = lane_xor<K - 1>(key);
    const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
    key = (li & (K >> 1)) == 0 ? mn : mx;
    return MergeSteps<K, (K >> 2)>::run(key, li);
}
// sort of one 64-bit key per lane inside aligned groups of G lanes (ascending)
template <int G> __device__ __forceinline__ unsigned long long group_sort(unsigned long long key, int li) {
    key = merge_level<2>(key, li);
    key = merge_level<4>(key, li);
    key = merge_level<8>(key, li);
    if (G >= 16)
        key = merge_level<16>(key, li);
    if (G >= 32)
        key = merge_level<32>(key, li);
    if (G >= 64)
        key = merge_level<64>(key, li);
    return key;
}

// Register-resident finishing for intervals with at most G pieces and no target overlaps or ties:
// in that case the MappedSegmentSet is just the pieces sorted by target start, every equivalence class of
// extractSegment (liftover/impl/halBlockMapper.cpp:348-352) has one member, no cut point is ever recorded
// (:385-387 needs a class of size > 1), so an output line is a maximal run of neighbours that
// canMergeRightWith (api/impl/halMappedSegment.cpp:109-161), and lines are then stably sorted by source start
// (liftover/impl/halLiftover.cpp:90).  One sub-wave group of G lanes per interval, one piece per lane.
static constexpr int FAST_LIST_CAP = 1024; // intervals per block of the G == 8 instantiation (its class lists live in LDS)

template <typename C, int G>
__global__ void __launch_bounds__(256) k_finish_fast(Mapped in, const uint32_t *__restrict__ offset, const uint32_t *__restrict__ count,
                                                     uint32_t nq, const int64_t *__restrict__ seqStart, int numSeq,
                                                     hgx_record *__restrict__ records, uint32_t *__restrict__ nOut,
                                                     uint32_t *__restrict__ generalList, unsigned long long *__restrict__ generalCount,
                                                     uint32_t *__restrict__ classLists, unsigned long long *__restrict__ classCounts) {
    // G == 8 looks at every interval: it finishes the ones with up to 8 pieces and sorts the others into three lists
    // (9-16, 17-32, 33-64 pieces; classLists + k*nq, classCounts[k]) that the wider instantiations then work through
    // densely.  The lists are gathered in LDS per block (one global atomic per block and class: a single counter word
    // takes ~90 atomics per microsecond); a block owns a contiguous range of intervals for that.
    constexpr int PER_WAVE = 64 / G;
    constexpr int CLS = G == 16 ? 0 : G == 32 ? 1 : 2;
    constexpr int LIST_CAP = FAST_LIST_CAP;
    __shared__ uint32_t sList[G == 8 ? 3 * LIST_CAP : 1];
    __shared__ uint32_t sCount[3];
    if (G == 8) {
        if (threadIdx.x < 3)
            sCount[threadIdx.x] = 0;
        __syncthreads();
    }
    const uint32_t nlist = G == 8 ? nq : (uint32_t)classCounts[CLS];
    const uint32_t *myList = classLists + (size_t)CLS * nq;
    const int lane = lane_id();
    const int li = lane & (G - 1);       // lane inside the group
    const int gbase = lane & ~(G - 1);   // first lane of the group
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << gbase;
    const unsigned long long INF = ~0ull;
    // G == 8: block b takes intervals [b * chunk, (b + 1) * chunk), its waves interleaved inside; else grid-stride over the list
    const uint32_t chunk = G == 8 ? (nq + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t first = G == 8 ? blockIdx.x * chunk + (threadIdx.x >> 6) * PER_WAVE
                                  : ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * PER_WAVE;
    const uint32_t limit = G == 8 ? (blockIdx.x * chunk + chunk < nq ? blockIdx.x * chunk + chunk : nq) : nlist;
    const uint32_t stride = G == 8 ? (blockDim.x >> 6) * PER_WAVE : ((gridDim.x * blockDim.x) >> 6) * PER_WAVE;
    // The loop is a chain of dependent memory round trips — list entry -> count and offset -> pieces -> (sequence start)
    // -> stores, and on this hardware the wait for a load also waits for every older store — that cost ~20 k cycles per
    // iteration whatever G (profiles/r01q_pmc.txt: a dozen reads in flight per CU).  It is therefore software-pipelined
    // three deep: while iteration i is worked on, the pieces of i+1, the count and offset of i+2 and the list entry of
    // i+3 are in flight, each requested before iteration i's stores so that no load ever waits for a store.
    typedef C P; // positions in the coordinate type of the alignment (32-bit arithmetic on narrow alignments)
    const int64_t ss0 = seqStart[0];
    struct Entry { // list stage
        uint32_t q;
        bool valid;
    };
    struct Head { // count / offset stage
        uint32_t q, base;
        int n;
        bool valid;
    };
    struct Pieces { // piece stage: one piece per lane
        uint32_t q, base;
        int n;
        bool gvalid;
        MappedRec r;
    };
    auto loadEntry = [&](uint32_t wb) {
        Entry e;
        const uint32_t slot = wb + (uint32_t)(lane / G);
        e.valid = wb < limit && slot < limit;
        e.q = e.valid ? (G == 8 ? slot : myList[slot]) : 0u;
        return e;
    };
    auto loadHead = [&](const Entry &e) {
        Head h;
        h.q = e.q;
        h.valid = e.valid;
        h.n = e.valid ? (int)count[e.q] : 0;
        h.base = e.valid ? offset[e.q] : 0u;
        return h;
    };
    // classifies the interval (G == 8: hands the larger ones to the lists) and requests its pieces
    auto loadPieces = [&](const Head &h) {
        Pieces c;
        c.q = h.q;
        c.base = h.base;
        c.n = h.n;
        c.gvalid = h.valid;
        if (h.valid && G == 8) {
            const uint32_t q = h.q;
            const int n = h.n;
            if (n == 0 && li == 0)
                nOut[q] = 0;
            if (n > 8 && li == 0) {
                if (n > 64) { // too many pieces for a wavefront: general path
                    const unsigned long long slot = atomicAdd(generalCount, 1ull);
                    generalList[slot] = q;
                } else {
                    const int cl = n <= 16 ? 0 : n <= 32 ? 1 : 2;
                    const uint32_t at = atomicAdd(&sCount[cl], 1u);
                    if (at < (uint32_t)LIST_CAP) {
                        sList[cl * LIST_CAP + at] = q;
                    } else { // the block's window is full (cannot happen with chunk <= LIST_CAP; kept as a guard)
                        const unsigned long long slot = atomicAdd(&classCounts[cl], 1ull);
                        classLists[(size_t)cl * nq + slot] = q;
                    }
                }
            }
            c.gvalid = n > 0 && n <= 8;
        }
        if (!c.gvalid)
            c.n = 0;
        c.r = MappedRec{};
        if (li < c.n)
            c.r = in.rec[c.base + li];
        return c;
    };
    Entry stageA = loadEntry(first);
    Head stageB = loadHead(stageA);
    stageA = loadEntry(first + stride);
    Pieces stageC = loadPieces(stageB);
    stageB = loadHead(stageA);
    stageA = loadEntry(first + 2 * stride);
    for (uint32_t wbase = first; wbase < limit; wbase += stride) {
        const Pieces cur = stageC;
        stageC = loadPieces(stageB);
        stageB = loadHead(stageA);
        stageA = loadEntry(wbase + 3 * stride);
        const bool gvalid = cur.gvalid;
        const uint32_t q = cur.q, base = cur.base;
        const int n = cur.n;
        if (!__any(gvalid))
            continue; // nothing for this instantiation among this wavefront's candidates
        const bool have = li < n;
        P tLo = 0, tHi = 0, sLo = 0, sHi = 0;
        uint8_t fl = 0;
        if (have) {
            const MappedRec &r = cur.r;
            tLo = (P)r.tLo;
            tHi = (P)(r.tLo + r.len - 1);
            sLo = (P)r.sLo;
            sHi = (P)(r.sLo + r.len - 1);
            fl = (uint8_t)r.flags;
        }
        // 1. order by target start
        unsigned long long key = have ? (((unsigned long long)(int64_t)tLo << 6) | (unsigned long long)lane) : INF;
        key = group_sort<G>(key, li);
        const int srcLane = (int)(key & 63ull);
        const bool occ = key != INF; // sorted position li holds a piece
        // the sorted key carries the target start; length, source start and flags come from the lane that loaded the piece
        const int32_t len = __shfl((int32_t)(tHi - tLo + 1), srcLane);
        tLo = (P)(key >> 6);
        tHi = (P)(tLo + len - 1);
        sLo = (P)shfl_pos<C>((int64_t)sLo, srcLane);
        sHi = (P)(sLo + len - 1);
        fl = (uint8_t)__shfl((int)fl, srcLane);
        int seq = 0;
        if (numSeq > 1 && occ) {
            int lo = 0, hi = numSeq;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (seqStart[mid] <= (int64_t)tLo)
                    lo = mid;
                else
                    hi = mid;
            }
            seq = lo;
        }
        // 2. neighbour relations (previous piece in target order)
        const P pTHi = lane_prev(tHi), pSLo = lane_prev(sLo), pSHi = lane_prev(sHi);
        const int pFl = lane_prev((int)fl), pSeq = lane_prev(seq);
        const bool hasPrev = occ && li > 0;
        const bool complex_ = hasPrev && tLo <= pTHi; // overlap or tie: needs the general algorithm
        bool mergePrev = false;
        if (hasPrev) {
            const bool sameStrands = (((int)fl ^ pFl) & (F_SREV | F_TREV)) == 0;
            const bool same = ((fl & F_SREV) != 0) == ((fl & F_TREV) != 0);
            const bool rOk = same ? (sLo - pSHi == 1) : (pSLo - sHi == 1);
            mergePrev = sameStrands && (tLo - pTHi == 1) && rOk && seq == pSeq;
        }
        const unsigned long long complexMask = __ballot(complex_) & gmask;
        const bool head = occ && !mergePrev;
        const unsigned long long headMask = __ballot(head) & gmask;
        if (gvalid && complexMask != 0) {
            if (li == 0) {
                const unsigned long long slot = atomicAdd(generalCount, 1ull);
                generalList[slot] = q;
            }
        }
        const bool doGroup = gvalid && complexMask == 0;
        // 3. a line runs from its head to the piece before the next head
        int back = li;
        if (head) {
            const unsigned long long above = (li + 1 < 64) ? ((headMask >> gbase) >> (li + 1)) : 0ull; // heads after me, group-relative
            const int nextHead = above ? li + 1 + (__ffsll((long long)above) - 1) : n;
            back = nextHead - 1;
        }
        const P bTHi = (P)shfl_pos<C>((int64_t)tHi, gbase + back), bSLo = (P)shfl_pos<C>((int64_t)sLo, gbase + back);
        const P lStart = tLo, lSrc = sLo < bSLo ? sLo : bSLo;
        const int64_t lEnd = (int64_t)bTHi + 1;
        // 4. stable sort of the lines by source start
        unsigned long long lkey = (head && doGroup) ? (((unsigned long long)(int64_t)lSrc << 6) | (unsigned long long)lane) : INF;
        lkey = group_sort<G>(lkey, li);
        const int lLane = (int)(lkey & 63ull);
        const bool isLine = lkey != INF;
        const int64_t oStart = shfl_pos<C>((int64_t)lStart, lLane), oEnd = shfl_pos<C>(lEnd - 1, lLane) + 1;
        const int64_t oSrc = (int64_t)(lkey >> 6); // (the sorted key carries the line's source start)
        const int oSeq = numSeq > 1 ? __shfl(seq, lLane) : 0, oFl = __shfl((int)fl, lLane);
        if (isLine) {
            hgx_record r;
            const int64_t ss = numSeq > 1 ? seqStart[oSeq] : ss0;
            r.query = q;
            r.tgt_start = oStart - ss;
            r.tgt_end = oEnd - ss;
            r.src_start = oSrc;
            r.tgt_seq = oSeq;
            r.strand = (oFl & F_DOT) ? '.' : ((oFl & F_TREV) ? '-' : '+');
            r.tgt_reversed = (oFl & F_TREV) ? 1 : 0;
            r._pad[0] = r._pad[1] = 0;
            records[base + li] = r;
        }
        if (doGroup && li == 0)
            nOut[q] = (uint32_t)__popcll(headMask);
    }
    if (G == 8) {
        __syncthreads();
        __shared__ unsigned long long sBase[3];
        if (threadIdx.x < 3) {
            const uint32_t c = sCount[threadIdx.x] < (uint32_t)LIST_CAP ? sCount[threadIdx.x] : (uint32_t)LIST_CAP;
            sBase[threadIdx.x] = c ? atomicAdd(&classCounts[threadIdx.x], (unsigned long long)c) : 0ull;
        }
        __syncthreads();
        for (int c = 0; c < 3; ++c) {
            const uint32_t cnt = sCount[c] < (uint32_t)LIST_CAP ? sCount[c] : (uint32_t)LIST_CAP;
            for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x)
                classLists[(size_t)c * nq + sBase[c] + k] = sList[c * LIST_CAP + k];
        }
    }
}

// single-pass runs (hgx_lift_kernels.hpp) keep the number of output lines per 64 intervals (a wavefront's worth); an
// interval finished after the counting launch adds its lines here
static constexpr int LIFT_TILE_SHIFT = 8;
// (otherLines: the statistics word for lines that do not come from merged records, one of its copies)
__device__ __forceinline__ void lift_add_late_lines(uint32_t *waveTotal, unsigned long long *otherLines, uint32_t q, int nl) {
    if (waveTotal && nl > 0) {
        atomicAdd(&waveTotal[q >> 6], (uint32_t)nl);
        atomicAdd(otherLines + (size_t)(blockIdx.x & (STAT_LINES - 1)) * STAT_PITCH, (unsigned long long)nl);
    }
}

// LDS-staged variant: capacity CAP pieces per query.
template <typename C, int CAP>
__global__ void __launch_bounds__(64) k_finish_lds(Mapped in, const uint32_t *__restrict__ offset, const uint32_t *__restrict__ count,
                                                   const uint32_t *__restrict__ qlist, const unsigned long long *__restrict__ qcount,
                                                   const int64_t *__restrict__ seqStart, int numSeq, hgx_record *__restrict__ records,
                                                   uint32_t *__restrict__ nOut, uint32_t *__restrict__ deferredList, uint32_t *__restrict__ needCap,
                                                   unsigned long long *counters, int blocks, uint32_t *waveTotal = nullptr,
                                                   unsigned long long *otherLines = nullptr) {
    // waveTotal (single-pass runs, hgx_lift_kernels.hpp): the interval's lines are added to the line count of its 64 intervals
    constexpr int CAP2 = 2 * CAP;
    __shared__ C s_tLo[CAP], s_tHi[CAP], s_sLo[CAP], s_sHi[CAP], s_tLo2[CAP], s_tHi2[CAP], s_sLo2[CAP], s_sHi2[CAP];
    __shared__ C s_bnd[CAP2], s_bnd2[CAP2], s_lStart[CAP], s_lEnd[CAP], s_lSrc[CAP], s_cut[32];
    __shared__ uint32_t s_ord[CAP2];
    __shared__ int32_t s_seq[CAP], s_lSeq[CAP];
    __shared__ uint8_t s_fl[CAP], s_fl2[CAP], s_alive[CAP], s_lStrand[CAP];
    const uint32_t nlist = (uint32_t)*qcount;
    for (uint32_t k = blockIdx.x; k < nlist; k += gridDim.x) {
        const uint32_t q = qlist[k];
        const int n = (int)count[q];
        FinishStore<C> S;
        S.tLo = s_tLo, S.tHi = s_tHi, S.sLo = s_sLo, S.sHi = s_sHi, S.fl = s_fl;
        S.tLo2 = s_tLo2, S.tHi2 = s_tHi2, S.sLo2 = s_sLo2, S.sHi2 = s_sHi2, S.fl2 = s_fl2;
        S.seq = s_seq, S.ord = s_ord, S.bnd = s_bnd, S.bnd2 = s_bnd2, S.alive = s_alive, S.cut = s_cut;
        S.lStart = s_lStart, S.lEnd = s_lEnd, S.lSrc = s_lSrc, S.lSeq = s_lSeq, S.lStrand = s_lStrand;
        S.cap = CAP, S.cap2 = CAP2, S.cutCap = 32;
        S.blocks = blocks;
        const uint32_t base = offset[q];
        int nl = finish_query<C, CAP / 64>(S, in, base, n, seqStart, numSeq);
        if (nl >= 0 && nl > n)
            nl = -nl; // records are written into the query's own slice of the grouped buffer (n slots)
        if (nl < 0) {
            if (threadIdx.x == 0) {
                const unsigned long long slot = atomicAdd(&counters[CNT_DEFERRED], 1ull);
                deferredList[slot] = q;
                needCap[q] = (uint32_t)(-nl);
                atomicMax(&counters[CNT_MAXNEED], (unsigned long long)(-nl));
                nOut[q] = 0;
            }
        } else {
            write_records(S, nl, (int32_t)q, records + base, seqStart);
            if (threadIdx.x == 0) {
                nOut[q] = (uint32_t)nl;
                lift_add_late_lines(waveTotal, otherLines, q, nl);
            }
        }
        wsync();
    }
}

// Global-scratch variant for deferred queries: slice k of `scratch` serves deferredList[k].
template <typename C>
__global__ void __launch_bounds__(64) k_finish_big(Mapped in, const uint32_t *__restrict__ offset, const uint32_t *__restrict__ count,
                                                   const uint32_t *__restrict__ deferredList, uint32_t nDeferred, int cap,
                                                   unsigned char *__restrict__ scratch, size_t sliceBytes,
                                                   const int64_t *__restrict__ seqStart, int numSeq, hgx_record *__restrict__ bigRecords,
                                                   uint32_t *__restrict__ nOut, unsigned long long *counters, int blocks,
                                                   int countOnDevice = 0, uint32_t *__restrict__ offsetOut = nullptr, uint32_t recordBase = 0,
                                                   uint32_t *waveTotal = nullptr, unsigned long long *otherLines = nullptr, int stageInLds = 0) {
    // stageInLds: the launch has sliceBytes of dynamic LDS and the interval is staged there instead of in its slice of the global
    // scratch (an interval of a few hundred pieces: the sorts and the sweeps are round trips to LDS, not to memory)
    extern __shared__ __attribute__((aligned(16))) unsigned char bigLds[];
    // countOnDevice (single-pass runs, hgx_lift_kernels.hpp): the number of deferred intervals is read from the counter block
    // (nDeferred = the slices the scratch area holds; more than that fails the run, the host grows the area and repeats it);
    // offsetOut: the interval's records are slice k of an area that starts recordBase records into the grouped buffer
    if (countOnDevice) {
        const unsigned long long have = counters[CNT_DEFERRED];
        if (have > (unsigned long long)nDeferred) {
            if (threadIdx.x == 0 && blockIdx.x == 0)
                counters[CNT_BIGFAIL] = 1;
        } else {
            nDeferred = (uint32_t)have;
        }
    }
    for (uint32_t k = blockIdx.x; k < nDeferred; k += gridDim.x) {
        const uint32_t q = deferredList[k];
        const int n = (int)count[q];
        unsigned char *p = stageInLds ? bigLds : scratch + (size_t)k * sliceBytes;
        auto carve = [&](size_t bytes) {
            unsigned char *r = p;
            p += (bytes + 15) & ~(size_t)15;
            return r;
        };
        const int cap2 = pow2_at_least(2 * cap);
        FinishStore<C> S;
        S.tLo = (C *)carve(sizeof(C) * cap), S.tHi = (C *)carve(sizeof(C) * cap), S.sLo = (C *)carve(sizeof(C) * cap),
        S.sHi = (C *)carve(sizeof(C) * cap);
        S.tLo2 = (C *)carve(sizeof(C) * cap), S.tHi2 = (C *)carve(sizeof(C) * cap), S.sLo2 = (C *)carve(sizeof(C) * cap),
        S.sHi2 = (C *)carve(sizeof(C) * cap);
        S.bnd = (C *)carve(sizeof(C) * cap2), S.bnd2 = (C *)carve(sizeof(C) * cap2);
        S.lStart = (C *)carve(sizeof(C) * cap), S.lEnd = (C *)carve(sizeof(C) * cap), S.lSrc = (C *)carve(sizeof(C) * cap);
        S.cut = (C *)carve(sizeof(C) * cap);
        S.ord = (uint32_t *)carve(4 * (size_t)cap2);
        S.seq = (int32_t *)carve(4 * (size_t)cap), S.lSeq = (int32_t *)carve(4 * (size_t)cap);
        S.fl = carve(cap), S.fl2 = carve(cap), S.alive = carve(cap), S.lStrand = carve(cap);
        S.cap = cap, S.cap2 = cap2, S.cutCap = cap;
        S.blocks = blocks;
        __threadfence_block();
        const int nl = finish_query<C, 8>(S, in, offset[q], n, seqStart, numSeq);
        if (nl < 0) {
            if (threadIdx.x == 0) {
                counters[CNT_BIGFAIL] = 1;
                atomicMax(&counters[CNT_MAXNEED], (unsigned long long)(-nl));
                nOut[q] = 0;
            }
        } else {
            write_records(S, nl, (int32_t)q, bigRecords + (size_t)k * cap, seqStart);
            if (threadIdx.x == 0) {
                nOut[q] = (uint32_t)nl;
                lift_add_late_lines(waveTotal, otherLines, q, nl);
                if (offsetOut)
                    offsetOut[q] = recordBase + k * (uint32_t)cap;
            }
        }
        wsync();
    }
}

// blocks mode: every interval with pieces goes to the general kernel
static __global__ void __launch_bounds__(256) k_all_general(const uint32_t *__restrict__ count, uint32_t nq, uint32_t *__restrict__ nOut,
                                                            uint32_t *__restrict__ generalList, unsigned long long *__restrict__ generalCount) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        if (count[q] == 0)
            nOut[q] = 0;
        else
            generalList[atomicAdd(generalCount, 1ull)] = q;
    }
}

// bytes of scratch per deferred query for capacity cap (must match the carving above)
template <typename C> inline size_t finishSliceBytes(int cap) {
    int cap2 = 1;
    while (cap2 < 2 * cap)
        cap2 <<= 1;
    auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
    return 8 * r(sizeof(C) * cap) + 2 * r(sizeof(C) * cap2) + 3 * r(sizeof(C) * cap) + r(sizeof(C) * cap) + r(4 * (size_t)cap2) +
           2 * r(4 * (size_t)cap) + 4 * r(cap);
}

// dense output: out[outOffset[q] + k] = records of q (from its grouped slice, or from the big-records area)
__global__ void __launch_bounds__(256) k_compact_records(const hgx_record *__restrict__ grouped, const uint32_t *__restrict__ offset,
                                                         const hgx_record *__restrict__ big, const int32_t *__restrict__ bigSlot, int bigCap,
                                                         const uint32_t *__restrict__ nOut, const uint32_t *__restrict__ outOffset, uint32_t nq,
                                                         hgx_record *__restrict__ out) {
    // eight lanes per query: most intervals produce a handful of records
    const uint32_t groupsTotal = (gridDim.x * blockDim.x) >> 3;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const uint32_t li = threadIdx.x & 7;
    for (uint32_t q = group; q < nq; q += groupsTotal) {
        const uint32_t n = nOut[q];
        if (n == 0)
            continue;
        const int32_t bs = bigSlot ? bigSlot[q] : -1;
        const hgx_record *src = bs >= 0 ? big + (size_t)bs * bigCap : grouped + offset[q];
        hgx_record *dst = out + outOffset[q];
        for (uint32_t k = li; k < n; k += 8)
            dst[k] = src[k];
    }
}

// hgx_record (40 B) -> the 20-byte wire form of the multi-GPU exchange (hal_amd/shard.py: pack_records): query, tgt_start,
// tgt_end, src_start as int32, then tgt_seq << 16 | strand << 8 | tgt_reversed
static __global__ void __launch_bounds__(256) k_pack_records(const hgx_record *__restrict__ in, uint32_t n, int32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const hgx_record r = in[i];
        int32_t *o = out + (size_t)i * 5;
        o[0] = (int32_t)r.query;
        o[1] = (int32_t)r.tgt_start;
        o[2] = (int32_t)r.tgt_end;
        o[3] = (int32_t)r.src_start;
        o[4] = (int32_t)(((uint32_t)r.tgt_seq << 16) | ((uint32_t)(uint8_t)r.strand << 8) | (uint32_t)r.tgt_reversed);
    }
}

// hgx_record (40 B) -> the 12-byte wire form: tgt_start, src_start (uint32) and (tgt_end - tgt_start) | tgt_seq << 22 |
// strand code << 29 | tgt_reversed << 31; the query index is not sent, a uint16 record count per interval is
// (k_wire12_counts).  *bad is set when a field does not fit (the caller then sends the 20-byte form).
static __global__ void __launch_bounds__(256) k_wire12_records(const hgx_record *__restrict__ in, uint32_t n, uint32_t *__restrict__ out,
                                                               unsigned int *bad) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const hgx_record r = in[i];
        const int64_t len = r.tgt_end - r.tgt_start;
        const uint32_t sc = r.strand == '+' ? 0u : r.strand == '-' ? 1u : r.strand == '.' ? 2u : 3u;
        if (len < 0 || len >= (1 << 22) || (uint64_t)r.tgt_start >> 32 || (uint64_t)r.src_start >> 32 || (uint32_t)r.tgt_seq >= 128u || sc == 3u ||
            r.tgt_reversed > 1)
            *bad = 1;
        uint32_t *o = out + (size_t)i * 3;
        o[0] = (uint32_t)r.tgt_start;
        o[1] = (uint32_t)r.src_start;
        o[2] = (uint32_t)len | ((uint32_t)r.tgt_seq << 22) | (sc << 29) | ((uint32_t)r.tgt_reversed << 31);
    }
}
// the 8-byte form for a receiver that only prints BED lines (halLiftover's output has no source coordinates: liftover/impl/
// halLiftover.cpp:94-106): tgt_start as uint32 and the third word of the 12-byte form; same limits, same flag
static __global__ void __launch_bounds__(256) k_wire8_records(const hgx_record *__restrict__ in, uint32_t n, uint32_t *__restrict__ out,
                                                              unsigned int *bad) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const hgx_record r = in[i];
        const int64_t len = r.tgt_end - r.tgt_start;
        const uint32_t sc = r.strand == '+' ? 0u : r.strand == '-' ? 1u : r.strand == '.' ? 2u : 3u;
        if (len < 0 || len >= (1 << 22) || (uint64_t)r.tgt_start >> 32 || (uint32_t)r.tgt_seq >= 128u || sc == 3u || r.tgt_reversed > 1)
            *bad = 1;
        out[2 * (size_t)i] = (uint32_t)r.tgt_start;
        out[2 * (size_t)i + 1] = (uint32_t)len | ((uint32_t)r.tgt_seq << 22) | (sc << 29) | ((uint32_t)r.tgt_reversed << 31);
    }
}
static __global__ void __launch_bounds__(256) k_wire12_counts(const uint32_t *__restrict__ nOut, uint32_t nq, uint32_t nqPadded,
                                                              uint16_t *__restrict__ out, unsigned int *bad) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nqPadded; q += gridDim.x * blockDim.x) {
        const uint32_t c = q < nq ? nOut[q] : 0u; // (the padding up to a multiple of 8 bytes is zero)
        if (c >> 16)
            *bad = 1;
        out[q] = (uint16_t)c;
    }
}
struct WireHeader { // include/hgx.h: hgx_liftover_wire_blob
    char magic[4];
    uint32_t format;
    int64_t firstQuery;
    uint64_t nq, nrec;
};
static_assert(sizeof(WireHeader) == 32, "blob header");
static __global__ void k_wire_header(WireHeader h, WireHeader *out) {
    *out = h;
}

// ---------------------------------------------------------------------------------------------
// The general algorithm of finish_query for intervals of at most 64 pieces, entirely in registers: one wavefront per
// interval, one piece per lane.  finish_query's LDS version spends its time in chains of dependent LDS reads issued by one
// lane (15-45 us per interval whatever its size); here
//   * the two sorts are ranks by counting (every lane compares itself with piece j, read with v_readlane, j = 0..n-1);
//   * overlap breaking is the common refinement computed from the ranks of the pieces' boundaries among the distinct
//     boundary values: piece i becomes rank(tHi+1) - rank(tLo) pieces, the k-th of them from boundary rank(tLo)+k to the next
//     (insertAndBreakOverlaps, api/impl/halSegmentMapper.cpp:475-520; the sorted boundaries are the only thing kept in LDS);
//   * the sequential part — BlockMapper::extractSegment over the set in target order (liftover/impl/halBlockMapper.cpp:
//     331-394: equivalence classes of equal target start, canMergeRightWith, cut points, erasure) — runs as scalar code that
//     reads the pieces with v_readlane; the set membership is one 64-bit mask.
// Intervals whose refined set has more than 64 members, or more than 64 cut points, are passed on (return false).
template <typename C> __device__ __forceinline__ C wave_read(C v, int j);
template <> __device__ __forceinline__ int32_t wave_read<int32_t>(int32_t v, int j) {
    return __builtin_amdgcn_readlane(v, j);
}
template <> __device__ __forceinline__ int64_t wave_read<int64_t>(int64_t v, int j) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), j);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <> __device__ __forceinline__ unsigned long long wave_read<unsigned long long>(unsigned long long v, int j) {
    return (unsigned long long)wave_read<int64_t>((int64_t)v, j);
}
template <typename C> __device__ __forceinline__ C wave_push(C v, int dstLane);
template <> __device__ __forceinline__ int32_t wave_push<int32_t>(int32_t v, int dstLane) {
    return __builtin_amdgcn_ds_permute(dstLane << 2, v);
}
template <> __device__ __forceinline__ int64_t wave_push<int64_t>(int64_t v, int dstLane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_permute(dstLane << 2, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_permute(dstLane << 2, (int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <typename C> __device__ __forceinline__ C wave_pull(C v, int srcLane);
template <> __device__ __forceinline__ int32_t wave_pull<int32_t>(int32_t v, int srcLane) {
    return __shfl(v, srcLane);
}
template <> __device__ __forceinline__ int64_t wave_pull<int64_t>(int64_t v, int srcLane) {
    return __shfl(v, srcLane);
}

// BlockMapper::extractSegment (liftover/impl/halBlockMapper.cpp:331-394) over a set of up to 64 * REGS members in target order — the
// formulation of finish_wave (below) for sets of more than 64 members.  The serial loop of finish_query, one lane over the
// staging arrays, spends an LDS round trip on every coordinate it looks at and looks at ten per pair of members: an interval of a
// hundred pieces took 120 us, 100 of them there.  Here everything that looks at coordinates is done for all members at once —
// where the classes of equal target start begin (gb), and for every member a which members of the class BEHIND its own it could
// be merged with if no cut point forbade it (canMergeRightWith, halMappedSegment.cpp:109-161: a 64-bit mask relative to that
// class's first member nb[a]; classes of more than 64 members go back to the serial loop) — and the sequential part (which
// classes a line swallows, what is erased, where the cut points are) runs as wave-uniform scalar code on bit sets of REGS words:
// the members still in the set, the class starts, the members whose target end is a cut point.  Same walk, same erasures, same
// lines as the serial loop, which remains the statement of what this computes; it has room for 32 cut points and passes an
// interval with more on to a larger staging area, this form has no such limit and gives the answer the larger area would.
// (explicit members and compile-time selection: an array indexed through a loop, even an unrollable one inside a lambda, went to
// scratch memory — and everything read back from there counts as divergent, which turns every v_readlane into a waterfall loop)
template <typename T, int N> struct RegFile {
    T r0, r1, r2, r3, r4, r5, r6, r7;
    template <int K> __device__ __forceinline__ T &ref() {
        static_assert(K < 8, "eight registers");
        if constexpr (K == 0) return r0;
        else if constexpr (K == 1) return r1;
        else if constexpr (K == 2) return r2;
        else if constexpr (K == 3) return r3;
        else if constexpr (K == 4) return r4;
        else if constexpr (K == 5) return r5;
        else if constexpr (K == 6) return r6;
        else return r7;
    }
    template <int K> __device__ __forceinline__ T get() const {
        return const_cast<RegFile *>(this)->template ref<K>();
    }
};
// a value per member, member j in lane j & 63 of register j >> 6; at(j) for a wave-uniform j
template <typename T, int REGS> struct RegSet {
    RegFile<T, REGS> f;
    template <int K = 1> __device__ __forceinline__ T pick(T r, int hi, int lo) const {
        if constexpr (K < REGS) {
            if (hi == K)
                r = wave_read<T>(f.template get<K>(), lo);
            return pick<K + 1>(r, hi, lo);
        } else {
            return r;
        }
    }
    __device__ __forceinline__ T at(int j) const {
        const int lo = __builtin_amdgcn_readfirstlane(j & 63), hi = __builtin_amdgcn_readfirstlane(j >> 6);
        return pick<1>(wave_read<T>(f.template get<0>(), lo), hi, lo);
    }
};
// a set of members, W words of 64 (bits at or behind n are never set)
template <int W> struct BitSet {
    RegFile<unsigned long long, W> f;
    template <int K = 0> __device__ __forceinline__ void zero() {
        if constexpr (K < W) {
            f.template ref<K>() = 0ull;
            zero<K + 1>();
        }
    }
    template <int K = 1> __device__ __forceinline__ unsigned long long pick(unsigned long long w, int k) const {
        if constexpr (K < W) {
            if (k == K)
                w = f.template get<K>();
            return pick<K + 1>(w, k);
        } else {
            return w;
        }
    }
    __device__ __forceinline__ unsigned long long word(int k) const { // (0 behind the last word)
        return k >= W ? 0ull : pick<1>(f.template get<0>(), k);
    }
    __device__ __forceinline__ bool get(int i) const {
        return ((word(i >> 6) >> (i & 63)) & 1ull) != 0;
    }
    template <int K = 0> __device__ __forceinline__ void clear(int i) {
        if constexpr (K < W) {
            if ((i >> 6) == K)
                f.template ref<K>() &= ~(1ull << (i & 63));
            clear<K + 1>(i);
        }
    }
    template <int K = 0> __device__ __forceinline__ void setSpan(int lo, int hi) { // members lo .. hi - 1
        if constexpr (K < W) {
            const int a = lo - 64 * K < 0 ? 0 : lo - 64 * K, b = hi - 64 * K > 64 ? 64 : hi - 64 * K;
            if (a < b)
                f.template ref<K>() |= (b >= 64 ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull);
            setSpan<K + 1>(lo, hi);
        }
    }
    __device__ __forceinline__ int firstFrom(int i, int n) const { // the first member at or behind i, or n
        if (i >= n)
            return n;
        int k = i >> 6;
        unsigned long long w = word(k) & (~0ull << (i & 63));
        while (!w && ++k < W)
            w = word(k);
        return w ? 64 * k + (int)__builtin_ctzll(w) : n;
    }
    __device__ __forceinline__ int lastAtOrBelow(int i) const { // the last member at or in front of i (0 if there is none)
        int k = i >> 6;
        unsigned long long w = word(k) & ((i & 63) == 63 ? ~0ull : ((1ull << ((i & 63) + 1)) - 1ull));
        while (!w && k > 0)
            w = word(--k);
        return w ? 64 * k + 63 - (int)__builtin_clzll(w) : 0;
    }
    __device__ __forceinline__ unsigned long long window(int i) const { // members i .. i + 63 as bits 0 .. 63
        const int k = i >> 6, sft = i & 63;
        unsigned long long w = word(k) >> sft;
        if (sft)
            w |= word(k + 1) << (64 - sft);
        return w;
    }
};
__device__ __forceinline__ unsigned long long low_bits(int len) { // len in 0 .. 64
    return len >= 64 ? ~0ull : ((1ull << len) - 1ull);
}
// The sorts of finish_query by ranks counted in registers: every lane holds its members (member j in lane j & 63 of register
// j >> 6), piece j's key is read by all lanes at once with v_readlane, j = 0 .. n - 1, and every lane counts the pieces in front
// of each of its own (equal keys: by index).  A hundred pieces: a few thousand instructions and one barrier, where the bitonic
// network in LDS took 36 stages of LDS round trips with a barrier each (40 k cycles an interval).
template <typename C, int REGS> struct RankSort {
    RegFile<C, REGS> tLo, tHi, sLo, sHi;
    RegFile<int32_t, REGS> rank;
    template <int K = 0> __device__ __forceinline__ void load(const FinishStore<C> &S, int lane, int n) {
        if constexpr (K < REGS) {
            const int i = lane + 64 * K;
            const bool in = i < n;
            tLo.template ref<K>() = in ? S.tLo[i] : (C)0;
            tHi.template ref<K>() = in ? S.tHi[i] : (C)0;
            sLo.template ref<K>() = in ? S.sLo[i] : (C)0;
            sHi.template ref<K>() = in ? S.sHi[i] : (C)0;
            rank.template ref<K>() = 0;
            load<K + 1>(S, lane, n);
        }
    }
    template <int M = 0> __device__ __forceinline__ void count(C a, C b, C c, C d, int j, int lane, int n) {
        if constexpr (M < REGS) {
            if (64 * M < n) { // (wave-uniform)
                const C mt = tLo.template get<M>(), mh = tHi.template get<M>(), ms = sLo.template get<M>(), me = sHi.template get<M>();
                const bool less = a != mt ? a < mt : b != mh ? b < mh : c != ms ? c < ms : d != me ? d < me : j < lane + 64 * M;
                rank.template ref<M>() += less ? 1 : 0;
            }
            count<M + 1>(a, b, c, d, j, lane, n);
        }
    }
    template <int K = 0> __device__ __forceinline__ void sweep(int lane, int n) {
        if constexpr (K < REGS) {
            const int lim = n - 64 * K < 64 ? n - 64 * K : 64;
            for (int jj = 0; jj < lim; ++jj)
                count<0>(wave_read<C>(tLo.template get<K>(), jj), wave_read<C>(tHi.template get<K>(), jj), wave_read<C>(sLo.template get<K>(), jj),
                         wave_read<C>(sHi.template get<K>(), jj), 64 * K + jj, lane, n);
            sweep<K + 1>(lane, n);
        }
    }
    template <int K = 0> __device__ __forceinline__ void store(FinishStore<C> &S, int lane, int n) {
        if constexpr (K < REGS) {
            const int i = lane + 64 * K;
            if (i < n) {
                const int r = rank.template get<K>();
                S.tLo2[r] = tLo.template get<K>();
                S.tHi2[r] = tHi.template get<K>();
                S.sLo2[r] = sLo.template get<K>();
                S.sHi2[r] = sHi.template get<K>();
                S.fl2[r] = S.fl[i];
            }
            store<K + 1>(S, lane, n);
        }
    }
};
template <typename C, int REGS> __device__ __forceinline__ void rank_sort_regs(FinishStore<C> &S, int n) {
    const int lane = lane_id();
    RankSort<C, REGS> R;
    R.load(S, lane, n);
    R.sweep(lane, n);
    R.store(S, lane, n);
    wsync();
    C *t;
    uint8_t *u;
    t = S.tLo, S.tLo = S.tLo2, S.tLo2 = t;
    t = S.tHi, S.tHi = S.tHi2, S.tHi2 = t;
    t = S.sLo, S.sLo = S.sLo2, S.sLo2 = t;
    t = S.sHi, S.sHi = S.sHi2, S.sHi2 = t;
    u = S.fl, S.fl = S.fl2, S.fl2 = u;
}
// the lines' order: stable by source start, S.ord[rank] = line
template <typename C, int REGS> struct RankLines {
    RegFile<C, REGS> src;
    RegFile<int32_t, REGS> rank;
    template <int K = 0> __device__ __forceinline__ void load(const FinishStore<C> &S, int lane, int nl) {
        if constexpr (K < REGS) {
            src.template ref<K>() = lane + 64 * K < nl ? S.lSrc[lane + 64 * K] : (C)0;
            rank.template ref<K>() = 0;
            load<K + 1>(S, lane, nl);
        }
    }
    template <int M = 0> __device__ __forceinline__ void count(C a, int j, int lane, int nl) {
        if constexpr (M < REGS) {
            if (64 * M < nl) {
                const C mine = src.template get<M>();
                rank.template ref<M>() += (a != mine ? a < mine : j < lane + 64 * M) ? 1 : 0;
            }
            count<M + 1>(a, j, lane, nl);
        }
    }
    template <int K = 0> __device__ __forceinline__ void sweep(int lane, int nl) {
        if constexpr (K < REGS) {
            const int lim = nl - 64 * K < 64 ? nl - 64 * K : 64;
            for (int jj = 0; jj < lim; ++jj)
                count<0>(wave_read<C>(src.template get<K>(), jj), 64 * K + jj, lane, nl);
            sweep<K + 1>(lane, nl);
        }
    }
    template <int K = 0> __device__ __forceinline__ void store(FinishStore<C> &S, int lane, int nl) {
        if constexpr (K < REGS) {
            if (lane + 64 * K < nl)
                S.ord[rank.template get<K>()] = (uint32_t)(lane + 64 * K);
            store<K + 1>(S, lane, nl);
        }
    }
};
template <typename C, int REGS> __device__ __forceinline__ void rank_lines_regs(FinishStore<C> &S, int nl) {
    const int lane = lane_id();
    RankLines<C, REGS> R;
    R.load(S, lane, nl);
    R.sweep(lane, nl);
    R.store(S, lane, nl);
    wsync();
}
template <typename C, int REGS> struct ExtractPrep { // the parallel part: one member per lane and register
    BitSet<REGS> gb;
    RegSet<unsigned long long, REGS> mergeRel;
    RegSet<int32_t, REGS> nb;
    bool tooWide = false;
    template <int K = 0> __device__ __forceinline__ void starts(const FinishStore<C> &S, int lane, int n) {
        if constexpr (K < REGS) {
            const int j = lane + 64 * K;
            gb.f.template ref<K>() = __ballot(j < n && (j == 0 || S.tLo[j] != S.tLo[j - 1]));
            starts<K + 1>(S, lane, n);
        }
    }
    template <int K = 0> __device__ __forceinline__ void partners(const FinishStore<C> &S, int lane, int n) {
        if constexpr (K < REGS) {
            const int j = lane + 64 * K;
            unsigned long long mr = 0;
            int first = n;
            if (j < n) {
                first = gb.firstFrom(j + 1, n); // the class behind this member's: [first, first + gsz)
                const int gsz = gb.firstFrom(first + 1, n) - first;
                if (gsz > 64 || (gb.get(j) && first - j > 64))
                    tooWide = true;
                const C tHi = S.tHi[j], sLo = S.sLo[j], sHi = S.sHi[j];
                const int fl = (int)S.fl[j], seq = S.seq[j];
                const bool same = ((fl & F_SREV) != 0) == ((fl & F_TREV) != 0);
                for (int c = 0; c < gsz && c < 64; ++c) {
                    const int b = first + c;
                    const int bFl = (int)S.fl[b];
                    if (S.seq[b] == seq && ((fl ^ bFl) & (F_SREV | F_TREV)) == 0 && S.tLo[b] - tHi == 1 &&
                        (same ? S.sLo[b] - sHi == 1 : sLo - S.sHi[b] == 1))
                        mr |= 1ull << c;
                }
            }
            mergeRel.f.template ref<K>() = mr;
            nb.f.template ref<K>() = first;
            partners<K + 1>(S, lane, n);
        }
    }
    template <int K = 0> __device__ __forceinline__ int classes() const {
        if constexpr (K < REGS)
            return (int)__popcll(gb.f.template get<K>()) + classes<K + 1>();
        else
            return 0;
    }
    __device__ __forceinline__ bool allSingle(int n) const {
        return classes() == n;
    }
    // (all classes single) line starts: member 0 and every member whose left neighbour cannot be merged with it
    template <int K = 0> __device__ __forceinline__ void lineStarts(BitSet<REGS> &L, int lane, int n, unsigned long long carry) {
        if constexpr (K < REGS) {
            const int j = lane + 64 * K;
            const unsigned long long m0 = __ballot(j < n && (mergeRel.f.template get<K>() & 1ull) != 0); // mergeable with member j + 1
            const unsigned long long valid = low_bits(n - 64 * K < 0 ? 0 : (n - 64 * K > 64 ? 64 : n - 64 * K));
            L.f.template ref<K>() = ~((m0 << 1) | carry) & valid;
            lineStarts<K + 1>(L, lane, n, m0 >> 63);
        }
    }
    template <int K = 0> __device__ __forceinline__ void writeLines(const FinishStore<C> &S, const BitSet<REGS> &L, int lane, int n, int before) {
        if constexpr (K < REGS) {
            const int j = lane + 64 * K;
            const unsigned long long w = L.f.template get<K>();
            if (j < n && ((w >> lane) & 1ull)) {
                const int l = before + (int)__popcll(w & ((1ull << lane) - 1ull));
                const int fragBack = L.firstFrom(j + 1, n) - 1; // the member in front of the next line's first
                const C iT = S.tLo[j], fT = S.tLo[fragBack], iH = S.tHi[j], fH = S.tHi[fragBack], iS = S.sLo[j], fS = S.sLo[fragBack];
                const int iFl = (int)S.fl[j];
                S.lStart[l] = iT < fT ? iT : fT; // halBlockLiftover.cpp:82-105
                S.lEnd[l] = (iH > fH ? iH : fH) + 1;
                S.lSrc[l] = iS < fS ? iS : fS;
                S.lSeq[l] = S.seq[j];
                S.lStrand[l] = (uint8_t)(((iFl & F_DOT) ? '.' : ((iFl & F_TREV) ? '-' : '+')) | ((iFl & F_TREV) ? 0x80 : 0));
            }
            writeLines<K + 1>(S, L, lane, n, before + (int)__popcll(w));
        }
    }
    __device__ __forceinline__ void singles(const FinishStore<C> &S, int lane, int n, int &nl) {
        BitSet<REGS> L;
        lineStarts(L, lane, n, 0ull); // (carry 0: member 0 begins a line — bit 0 of ~(m0 << 1) is set)
        nl = 0;
        countLines(L, nl);
        writeLines(S, L, lane, n, 0);
    }
    template <int K = 0> __device__ __forceinline__ void countLines(const BitSet<REGS> &L, int &nl) const {
        if constexpr (K < REGS) {
            nl += (int)__popcll(L.f.template get<K>());
            countLines<K + 1>(L, nl);
        }
    }
};
template <typename C, int REGS> __device__ __forceinline__ void extract_segment_regs(FinishStore<C> &S, int n, int &nlOut, int &failOut) {
    const int lane = lane_id();
    ExtractPrep<C, REGS> P;
    P.starts(S, lane, n);
    P.partners(S, lane, n);
    nlOut = 0;
    failOut = 0;
    if (__any(P.tooWide)) { // a class of more than 64 members: the serial loop's
        failOut = 2;
        return;
    }
    if (P.allSingle(n)) {
        // Every class has one member (no two members with the same target range: the usual set).  Then a line swallows the member
        // behind its last one exactly when the two can be merged (there is no cut point without a class of two), what it erases lies
        // behind it, and the lines are the maximal runs of members each mergeable with its right neighbour: no loop at all — a
        // line begins at member 0 and behind every member that cannot be merged with the next one.
        P.singles(S, lane, n, nlOut);
        return;
    }
    BitSet<REGS> alive, cut;
    alive.zero();
    cut.zero();
    alive.setSpan(0, n);
    int nl = 0;
    for (int i = alive.firstFrom(0, n); i < n; i = alive.firstFrom(i + 1, n)) {
        i = __builtin_amdgcn_readfirstlane(i);
        const int ce = P.gb.firstFrom(i + 1, n);
        unsigned long long v1 = alive.window(i) & low_bits(ce - i); // what is left of i's class from i on, bit 0 = member v1base
        int v1base = i, v1n = (int)__popcll(v1);
        int nxt = alive.firstFrom(ce, n);
        int fragBack = i;
        while (nxt < n) {
            // the next v1n members of the class nxt is in, as far as it has them
            unsigned long long rest = alive.window(nxt) & low_bits(P.gb.firstFrom(nxt + 1, n) - nxt), v2 = 0;
            int v2n = 0, last = 0;
            while (rest && v2n < v1n) {
                last = (int)__builtin_ctzll(rest);
                v2 |= 1ull << last;
                rest &= rest - 1;
                ++v2n;
            }
            const int after = alive.firstFrom(nxt + last + 1, n);
            bool can = v1n == v2n;
            for (unsigned long long m1 = v1, m2 = v2; m1 && can; m1 &= m1 - 1, m2 &= m2 - 1) { // member by member
                const int a = v1base + (int)__builtin_ctzll(m1), b = nxt + (int)__builtin_ctzll(m2);
                const int rel = b - P.nb.at(a);
                can = rel >= 0 && rel < 64 && ((P.mergeRel.at(a) >> rel) & 1ull) != 0 && !cut.get(a);
            }
            if (!can)
                break;
            fragBack = nxt; // (the first of v2: nxt itself is in the set)
            alive.clear(fragBack); // erased from the set (halBlockMapper.cpp:389-391)
            v1 = v2;
            v1base = nxt;
            v1n = v2n;
            nxt = after;
        }
        if (v1n > 1) // a cut point at the end of the last class (:382-386): no line merges across it any more
            cut.setSpan(P.gb.lastAtOrBelow(fragBack), P.gb.firstFrom(fragBack + 1, n));
        if (lane == 0) { // halBlockLiftover.cpp:82-105
            const C iT = S.tLo[i], fT = S.tLo[fragBack], iH = S.tHi[i], fH = S.tHi[fragBack], iS = S.sLo[i], fS = S.sLo[fragBack];
            const int iFl = (int)S.fl[i];
            S.lStart[nl] = iT < fT ? iT : fT;
            S.lEnd[nl] = (iH > fH ? iH : fH) + 1;
            S.lSrc[nl] = iS < fS ? iS : fS;
            S.lSeq[nl] = S.seq[i];
            // strand character in the low 7 bits, the piece's own orientation in bit 7
            S.lStrand[nl] = (uint8_t)(((iFl & F_DOT) ? '.' : ((iFl & F_TREV) ? '-' : '+')) | ((iFl & F_TREV) ? 0x80 : 0));
        }
        ++nl;
    }
    nlOut = nl;
}

// -DHGX_LIFT_PROFILE (make profile-lib: hal_amd/libhgx_prof.so, loaded with HGX_LIB_PATH): lane 0 of every wavefront of
// k_lift_merged (PROFILE_KERNEL=1), of k_lift_classify (2) or inside finish_wave (3) adds up the shader cycles it spends in each
// phase; k_lift_totals prints the sums.
#ifdef HGX_LIFT_PROFILE
__device__ unsigned long long g_liftProfile[8192 * 4 * 8]; // [workgroup][wavefront][phase]: plain stores, summed by the epilogue
#define LIFT_PROF_DECL unsigned long long profT = __builtin_readcyclecounter(), profAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define LIFT_PROF(i)                                                                                                                         \
    {                                                                                                                                        \
        const unsigned long long now = __builtin_readcyclecounter();                                                                         \
        profAcc[i] += now - profT;                                                                                                           \
        profT = now;                                                                                                                         \
    }
#if HGX_LIFT_PROFILE == 3 // the phases of finish_wave only (slots 0..6): LIFT_PROF moves the clock, LIFT_PROF3 books
#define LIFT_PROF3(i)                                                                                                                        \
    {                                                                                                                                        \
        const unsigned long long now = __builtin_readcyclecounter();                                                                         \
        profAcc[i] += now - profT;                                                                                                           \
        profT = now;                                                                                                                         \
    }
#undef LIFT_PROF
#define LIFT_PROF(i) profT = __builtin_readcyclecounter();
#else
#define LIFT_PROF3(i)
#endif
#define LIFT_PROF_FLUSH                                                                                                                      \
    if (lane == 0 && blockIdx.x < 8192) {                                                                                                    \
        profAcc[7] = 1;                                                                                                                      \
        for (int i = 0; i < 8; ++i)                                                                                                          \
            g_liftProfile[(blockIdx.x * 4 + w) * 8 + i] = profAcc[i];                                                                        \
    }
#define LIFT_PROF_ARG , profAcc, profT
#define LIFT_PROF_PARAMS , unsigned long long *profAcc, unsigned long long &profT
#else
#define LIFT_PROF_ARG
#define LIFT_PROF_PARAMS
#define LIFT_PROF_DECL
#define LIFT_PROF(i)
#define LIFT_PROF3(i)
#define LIFT_PROF_FLUSH
#endif

template <typename C> struct WaveLines { // line l of the interval lives in lane l
    C lStart, lEnd, lSrc;
    int lSeq, lStrand; // strand character in the low 7 bits, the piece's own orientation in bit 7
    int rank;          // position of the line in the output (stable order by source start)
    int nl;
};

// the pieces of lanes 0..n-1 into MappedSegmentSet order: (target, then source) in forward coordinates
template <typename C> __device__ __forceinline__ void wave_sort_pieces(int lane, int n, C &tLo, C &tHi, C &sLo, C &sHi, int &fl) {
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const C jt = wave_read<C>(tLo, j), jh = wave_read<C>(tHi, j), js = wave_read<C>(sLo, j), jS = wave_read<C>(sHi, j);
        const bool less = jt != tLo ? jt < tLo : jh != tHi ? jh < tHi : js != sLo ? js < sLo : jS != sHi ? jS < sHi : j < lane;
        rank += less ? 1 : 0;
    }
    if (lane >= n)
        rank = lane;
    tLo = wave_push<C>(tLo, rank);
    tHi = wave_push<C>(tHi, rank);
    sLo = wave_push<C>(sLo, rank);
    sHi = wave_push<C>(sHi, rank);
    fl = __builtin_amdgcn_ds_permute(rank << 2, fl);
}

// n (wave-uniform, 1..64) pieces in lanes 0..n-1; sD: 128 coordinates of LDS, sOwn: 64 bytes of LDS, private to the wavefront
template <typename C>
__device__ __forceinline__ bool finish_wave(const int lane, int n, C tLo, C tHi, C sLo, C sHi, int fl, const int64_t *__restrict__ seqStart,
                                            const int numSeq, C *sD, uint8_t *sOwn, WaveLines<C> &L LIFT_PROF_PARAMS) {
    LIFT_PROF3(0) // (look-ups, records, compaction: everything in front of the algorithm)
    wave_sort_pieces<C>(lane, n, tLo, tHi, sLo, sHi, fl);
    LIFT_PROF3(1) // first sort
    {
        const C pTLo = lane_prev(tLo), pTHi = lane_prev(tHi);
        const bool hasPrev = lane > 0 && lane < n;
        const bool sameT = hasPrev && tLo == pTLo && tHi == pTHi;
        if (__any(hasPrev && tLo <= pTHi && !sameT)) {
            // ---- refinement: every piece cut at every boundary (a piece's start, a piece's end + 1) inside it ----
            const bool valid = lane < n;
            const C A = tLo, B = tHi + 1;
            bool firstA = valid, firstB = valid; // this lane holds the canonical occurrence of the value
            for (int j = 0; j < n; ++j) {
                const C aj = wave_read<C>(A, j), bj = wave_read<C>(B, j);
                if (j < lane && aj == A)
                    firstA = false;
                if (aj == B || (j < lane && bj == B))
                    firstB = false;
            }
            int rA = 0, rB = 0; // number of distinct boundary values below mine
            const int fa = firstA ? 1 : 0, fb = firstB ? 1 : 0;
            for (int j = 0; j < n; ++j) {
                const C aj = wave_read<C>(A, j), bj = wave_read<C>(B, j);
                const int faj = __builtin_amdgcn_readlane(fa, j), fbj = __builtin_amdgcn_readlane(fb, j);
                rA += (faj && aj < A ? 1 : 0) + (fbj && bj < A ? 1 : 0);
                rB += (faj && aj < B ? 1 : 0) + (fbj && bj < B ? 1 : 0);
            }
            if (firstA)
                sD[rA] = A;
            if (firstB)
                sD[rB] = B;
            const int c = valid ? rB - rA : 0; // pieces this one becomes
            int incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o)
                    incl += up;
            }
            const int m = __builtin_amdgcn_readlane(incl, 63);
            if (m > 64)
                return false;
            const int off = incl - c;
            sOwn[lane] = 0;
            wave_lds_fence();
            if (c > 0)
                sOwn[off] = (uint8_t)(lane + 1);
            wave_lds_fence();
            int mark = (int)sOwn[lane];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(mark, o);
                if (lane >= o && up > mark)
                    mark = up;
            }
            const int owner = mark > 0 ? mark - 1 : 0;
            const int k = lane - __shfl(off, owner);
            const int ra = __shfl(rA, owner);
            const C otLo = wave_pull<C>(tLo, owner), osLo = wave_pull<C>(sLo, owner), osHi = wave_pull<C>(sHi, owner);
            const int ofl = __shfl(fl, owner);
            if (lane < m) {
                const C lo = sD[ra + k], hi = sD[ra + k + 1] - 1;
                const bool opposite = ((ofl & F_SREV) != 0) != ((ofl & F_TREV) != 0);
                tLo = lo;
                tHi = hi;
                if (!opposite) { // the source side sliced in lock-step (MappedSegment::slice, halMappedSegment.cpp:395-402)
                    sLo = osLo + (lo - otLo);
                    sHi = osLo + (hi - otLo);
                } else {
                    sLo = osHi - (hi - otLo);
                    sHi = osHi - (lo - otLo);
                }
                fl = ofl;
            } else {
                tLo = tHi = sLo = sHi = 0;
                fl = 0;
            }
            wave_lds_fence();
            n = m;
            LIFT_PROF3(2) // refinement
            wave_sort_pieces<C>(lane, n, tLo, tHi, sLo, sHi, fl);
            LIFT_PROF3(3) // second sort
        }
    }
    {   // equal keys: the set keeps one
        const C pTLo = lane_prev(tLo), pTHi = lane_prev(tHi), pSLo = lane_prev(sLo), pSHi = lane_prev(sHi);
        const bool dup = lane > 0 && lane < n && tLo == pTLo && tHi == pTHi && sLo == pSLo && sHi == pSHi;
        if (__any(dup)) {
            const unsigned long long keep = __ballot(lane < n && !dup);
            // (the dropped lanes all aim at lane 63: with n <= 64 and at least one of them it is never a kept lane's place)
            const int dest = (lane < n && !dup) ? (int)__popcll(keep & ((1ull << lane) - 1ull)) : 63;
            tLo = wave_push<C>(tLo, dest);
            tHi = wave_push<C>(tHi, dest);
            sLo = wave_push<C>(sLo, dest);
            sHi = wave_push<C>(sHi, dest);
            fl = __builtin_amdgcn_ds_permute(dest << 2, fl);
            n = (int)__popcll(keep);
        }
    }
    int seq = 0;
    if (numSeq > 1 && lane < n) {
        int lo = 0, hi = numSeq;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seqStart[mid] <= (int64_t)tLo)
                lo = mid;
            else
                hi = mid;
        }
        seq = lo;
    }
    LIFT_PROF3(4) // duplicates, sequences
    // ---- extractSegment over the set in target order (finish_query has the LDS original) ----
    // After the refinement two members' target ranges are equal or disjoint, so "equal target start" (the classes the reference
    // compares, halBlockMapper.cpp:340-352), "equal target end" (a cut point is a target end, :382-386) and "same class" are the
    // same thing: a class is a run of neighbours, its members in source order.  Everything that looks at coordinates is done for
    // all members at once — where classes begin (gb), and for every member a which members b of the class behind its own it
    // could be merged with if no cut point forbade it (mergeable[a], canMergeRightWith: halMappedSegment.cpp:109-161) — and the
    // sequential part (which classes a line swallows, what is erased, where cut points are) is left with 64-bit masks.
    const unsigned long long all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const C prevTLo = lane_prev(tLo); // (a DPP move: made by all lanes, not behind a condition that switches its source lane off)
    const unsigned long long gb = __ballot(lane < n && (lane == 0 || tLo != prevTLo)); // first members of the classes
    unsigned long long mergeable = 0;
    {
        const unsigned long long rest = lane < 63 ? (gb & all) >> (lane + 1) : 0ull; // class starts behind this member
        int nb = n, gsz = 0;                                                           // the class behind this member's: [nb, nb + gsz)
        if (lane < n && rest) {
            const int z = __ffsll((long long)rest) - 1;
            nb = lane + 1 + z;
            const unsigned long long rest2 = z < 63 ? rest >> (z + 1) : 0ull;
            gsz = (rest2 ? nb + __ffsll((long long)rest2) : n) - nb;
        }
        const bool same = ((fl & F_SREV) != 0) == ((fl & F_TREV) != 0);
        for (int c = 0; __any(c < gsz); ++c) {
            const int b = (nb + c) & 63;
            const C bT = wave_pull<C>(tLo, b), bSL = wave_pull<C>(sLo, b), bSH = wave_pull<C>(sHi, b);
            const int bFl = __shfl(fl, b), bSeq = __shfl(seq, b);
            if (c < gsz && bSeq == seq && ((fl ^ bFl) & (F_SREV | F_TREV)) == 0 && bT - tHi == 1 && (same ? bSL - sHi == 1 : sLo - bSH == 1))
                mergeable |= 1ull << b;
        }
    }
    unsigned long long alive = all, cut = 0; // cut: the members whose target end is a cut point
    int nl = 0, lineFirst = 0, lineBack = 0; // line l's first member and the first member of the last class it swallowed: lane l
    auto firstFrom = [&](unsigned long long m, int i) -> int { // first member of m at or behind i, or n
        if (i >= n)
            return n;
        const unsigned long long mm = (m & all) >> i;
        return mm ? i + (__ffsll((long long)mm) - 1) : n;
    };
    auto span = [&](int lo, int hi) -> unsigned long long { // the members lo .. hi - 1
        const unsigned long long upTo = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
        return upTo & ~((1ull << lo) - 1ull);
    };
    for (int i = firstFrom(alive, 0); i < n; i = firstFrom(alive, i + 1)) {
        unsigned long long v1 = alive & span(i, firstFrom(gb, i + 1)); // what is left of i's class from i on
        int v1n = (int)__popcll(v1);
        int nxt = firstFrom(alive, firstFrom(gb, i + 1));
        int fragBack = i;
        while (nxt < n) {
            // the next v1n members of the class nxt is in, as far as it has them
            unsigned long long rest = alive & span(nxt, firstFrom(gb, nxt + 1)), v2 = 0;
            int v2n = 0, last = nxt;
            while (rest && v2n < v1n) {
                last = __ffsll((long long)rest) - 1;
                v2 |= 1ull << last;
                rest &= rest - 1;
                ++v2n;
            }
            const int after = firstFrom(alive, last + 1);
            bool can = v1n == v2n;
            for (unsigned long long m1 = v1, m2 = v2; m1 && can; m1 &= m1 - 1, m2 &= m2 - 1) { // member by member
                const int a = __ffsll((long long)m1) - 1, b = __ffsll((long long)m2) - 1;
                const unsigned long long ma = (unsigned long long)wave_read<int64_t>((int64_t)mergeable, a);
                can = ((ma >> b) & 1ull) != 0 && ((cut >> a) & 1ull) == 0;
            }
            if (!can)
                break;
            fragBack = __ffsll((long long)v2) - 1;
            alive &= ~(1ull << fragBack); // erased from the set (halBlockMapper.cpp:389-391)
            v1 = v2;
            v1n = v2n;
            nxt = after;
        }
        if (v1n > 1) { // a cut point at the end of the last class (:382-386): no line merges across it any more
            const int g0 = 63 - __clzll((long long)(gb & span(0, fragBack + 1)));
            cut |= span(g0, firstFrom(gb, fragBack + 1));
        }
        if (lane == nl) {
            lineFirst = i;
            lineBack = fragBack;
        }
        ++nl;
    }
    {   // halBlockLiftover.cpp:82-105
        const C ti = wave_pull<C>(tLo, lineFirst), tf = wave_pull<C>(tLo, lineBack), hi_ = wave_pull<C>(tHi, lineFirst),
                hf = wave_pull<C>(tHi, lineBack), si = wave_pull<C>(sLo, lineFirst), sf = wave_pull<C>(sLo, lineBack);
        const int fi = __shfl(fl, lineFirst), seqI = __shfl(seq, lineFirst);
        L.lStart = L.lEnd = L.lSrc = 0;
        L.lSeq = L.lStrand = 0;
        if (lane < nl) {
            L.lStart = ti < tf ? ti : tf;
            L.lEnd = (hi_ > hf ? hi_ : hf) + 1;
            L.lSrc = si < sf ? si : sf;
            L.lSeq = seqI;
            L.lStrand = ((fi & F_DOT) ? '.' : ((fi & F_TREV) ? '-' : '+')) | ((fi & F_TREV) ? 0x80 : 0);
        }
    }
    LIFT_PROF3(5) // extractSegment
    // stable sort of the lines by source start (halLiftover.cpp:90), again a rank by counting
    int lrank = 0;
    for (int j = 0; j < nl; ++j) {
        const C js = wave_read<C>(L.lSrc, j);
        lrank += (js < L.lSrc || (js == L.lSrc && j < lane)) ? 1 : 0;
    }
    L.rank = lrank;
    L.nl = nl;
    LIFT_PROF3(6) // line order
    return true;
}

// One general interval of a single-pass run (hgx_lift_kernels.hpp), by one wavefront, from the unmerged table of the whole path
// to finished records without a round trip of the pieces through HBM: the interval's records (at most 64 in reach;
// k_locate_through's rule for which they are) are clipped in the lanes that loaded them, compacted, finished by finish_wave and
// written to a slice of the grouped buffer reserved from the same segment counters k_locate_through appends through.
// Returns 1 with (nl, base) = where k_lift_merged finds the records, 0 when the interval has to go on to k_locate_through +
// k_finish_lds (more records in reach, or finish_wave passes it on), -1 when the grouped buffer is full (the host repeats the
// batch with larger buffers).  All arguments wave-uniform.
template <typename C> struct GeneralTable {
    const uint32_t *coarse, *starts;
    int shift;
    const ComposedRec<C> *recs;
    int64_t genomeLength;
    const int64_t *seqStart;
    int numSeq;
    int64_t seqStart0;   // seqStart[0] (known to the host: no load in front of the record store)
    hgx_record *records; // the grouped buffer
    uint32_t cap;
    unsigned long long *segCounters, *counters;
};
template <typename C>
__device__ __forceinline__ int general_interval(const int lane, const GeneralTable<C> &G, const uint32_t q, const int64_t gs, const int64_t ge,
                                                const uint8_t st, C *sD, uint8_t *sOwn, uint32_t &used, int &nlOut, uint32_t &baseOut LIFT_PROF_PARAMS) {
    const int64_t geIn = ge < G.genomeLength ? ge : G.genomeLength - 1;
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)G.coarse[gs >> G.shift]);
    const uint32_t kEnd = (uint32_t)__builtin_amdgcn_readfirstlane((int)G.starts[(geIn >> G.shift) + 1]);
    nlOut = 0;
    baseOut = 0;
    if (kEnd > k0 + 128u)
        return 0;
    // up to 128 records in reach (the buckets of the interval's ends hold records that do not touch it), two rounds of 64; the ones
    // that overlap the interval are clipped in the lanes that loaded them and packed into the lanes from 0 on — up to 64 of them
    C tLo = 0, tHi = 0, sLo = 0, sHi = 0;
    int fl = 0, n = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t round = 0; round < 2u && k0 + 64u * round < kEnd; ++round) { // (wave-uniform)
        C rtLo = 0, rtHi = 0, rsLo = 0, rsHi = 0;
        int rfl = 0;
        bool have = false;
        const uint32_t k = k0 + 64u * round + (uint32_t)lane;
        if (k < kEnd) {
            const ComposedRec<C> r = G.recs[k];
            const int64_t pLo = (int64_t)r.sLo, pHi = pLo + (int64_t)r.len - 1;
            if (pLo <= ge && pHi >= gs) { // clipped to the interval (k_locate_through)
                const int64_t c = pLo > gs ? pLo : gs, d = pHi < ge ? pHi : ge;
                const int64_t len = d - c + 1, delta = c - pLo;
                have = true;
                rsLo = (C)c;
                rsHi = (C)d;
                rtLo = (C)((int64_t)r.so + ((r.mEncF & 1u) ? (int64_t)r.len - delta - len : delta));
                rtHi = (C)((int64_t)rtLo + len - 1);
                rfl = (int)((((r.mEncF & 1u) ? F_TREV : 0u) | (st == '.' ? (uint32_t)F_DOT : 0u)) ^ (st == '-' ? (uint32_t)(F_SREV | F_TREV) : 0u));
            }
        }
        const unsigned long long hm = __ballot(have);
        const int m = (int)__popcll(hm);
        used += have ? 1u : 0u;
        if (m == 0)
            continue;
        if (n + m > 64)
            return 0; // more pieces than lanes: the LDS finishing kernel's
        // to lanes n .. n + m - 1 (the ones left out aim at a lane that is not among them: if there is none, nobody is left out)
        const int dump = n > 0 ? 0 : 63;
        const int dest = have ? n + (int)__popcll(hm & below) : dump;
        rtLo = wave_push<C>(rtLo, dest);
        rtHi = wave_push<C>(rtHi, dest);
        rsLo = wave_push<C>(rsLo, dest);
        rsHi = wave_push<C>(rsHi, dest);
        rfl = __builtin_amdgcn_ds_permute(dest << 2, rfl);
        if (lane >= n && lane < n + m) {
            tLo = rtLo;
            tHi = rtHi;
            sLo = rsLo;
            sHi = rsHi;
            fl = rfl;
        }
        n += m;
    }
    if (n == 0)
        return 1;
    WaveLines<C> L;
    L.nl = 0;
    LIFT_PROF(4) // table look-ups, records
    if (!finish_wave<C>(lane, n, tLo, tHi, sLo, sHi, fl, G.seqStart, G.numSeq, sD, sOwn, L LIFT_PROF_ARG))
        return 0;
    LIFT_PROF(5) // the algorithm
    // a slice of the grouped buffer for the records
    const uint32_t seg = blockIdx.x % NSEG, segCap = G.cap / NSEG;
    unsigned long long b = 0;
    if (lane == 0)
        b = atomicAdd(G.segCounters + (size_t)seg * SEG_PITCH, (unsigned long long)L.nl);
    b = __shfl(b, 0);
    if (b + (unsigned long long)L.nl > segCap) { // the retry sizes the buffers from the counters
        if (lane == 0)
            G.counters[CNT_OVERFLOW] = 1;
        return -1;
    }
    const uint32_t base = seg * segCap + (uint32_t)b;
    LIFT_PROF(6) // the reservation
    if (lane < L.nl) {
        hgx_record r;
        const int64_t ss = G.numSeq > 1 ? G.seqStart[L.lSeq] : G.seqStart0;
        r.query = (int64_t)q;
        r.tgt_start = (int64_t)L.lStart - ss;
        r.tgt_end = (int64_t)L.lEnd - ss;
        r.src_start = (int64_t)L.lSrc;
        r.tgt_seq = L.lSeq;
        r.strand = (char)(L.lStrand & 0x7F);
        r.tgt_reversed = (uint8_t)((L.lStrand >> 7) & 1);
        r._pad[0] = r._pad[1] = 0;
        G.records[base + (uint32_t)L.rank] = r;
    }
    nlOut = L.nl;
    baseOut = base;
    return 1;
}

} // namespace hgx
