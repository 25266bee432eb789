// hal2maf's device stage from piece closures (gfx950): the rows of the columns where the row STRUCTURE changes, and of no others.
//
// The reference walks the whole column tree again for every reference base (api/impl/halColumnIterator.cpp:246-355), and the
// first version of this stage did the same on a lane per column: two walks (count, emit) of each of config 3's 54.7 M columns,
// 5.7 GB of rows written to keep the 2.28 M that begin a run.  A column's rows are the rows of the column before it, each
// advanced by one base on its strand, unless one of the segments the walk passes through ends between the two columns — and the
// walk's segments are the top and bottom segments around every base of the column (parent, child and paralogy hops keep the
// offset inside segments of equal length; a parse step changes the tiling, not the base).  So:
//
//   where runs begin      two sweeps over the tree, a byte per base, like the depth sweeps (hgx_column_kernels.hpp): bottom-up
//                         D_x[q] = 1 when a segment boundary lies between bases q-1 and q of genome x anywhere in the tree below x
//                         (x's own bottom segments, and the D of every top segment of every child that hangs under the bottom
//                         segment, read in the child's orientation); top-down along the path from the top of the scope to the
//                         reference F_x[q] = the same for the whole column of q (its topmost ancestor's D and the top segments'
//                         boundaries on the way).  F of the reference marks a superset of the columns that begin a run;
//   how many rows         the depth sweeps with sums (--countDupes without the -1): S_x[q] = the reported bases in the tree below
//                         base q, A_ref[p] = S at p's topmost ancestor = the rows of column p;
//   the rows              with every subtree's size known, the k-th row of a column — the reference's insertion order: the
//                         reference base, its ancestors upwards, then what hangs off the path from the top down, the reference's
//                         own paralogs and children last (recursiveUpdate's order, :246-355, :556-744) — is found by walking up
//                         the path once and down ONE branch: no stack, no second pass, a lane per row (k_maf_rows);
//   the heads             a marked column whose rows are its predecessor's advanced (adjacent segments mapped to adjacent
//                         places) is dropped by comparing it with the marked column before it (k_maf_heads).
#pragma once
#include "hgx_column_kernels.hpp"
#include "hgx_scan_kernels.hpp"

namespace hgx {

// ---- where runs begin ----
struct BreakChild {
    const int32_t *enc;   // the parent's child link array for this slot
    const void *top;      // the child's TopRec table
    const uint8_t *track; // the child's D (children without bottom segments in scope have none and are left out)
};
struct BreakChildren {
    BreakChild c[SWEEP_MAX_CHILDREN];
    int n;
    int noRing; // --noDupes (SweepChildren::noRing)
};
// eight bases of a track as one unaligned word; `rev`: the bases at the mirrored place, back to front
HGX_DEV __forceinline__ unsigned long long break_word(const uint8_t *p, bool rev) {
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);
    return rev ? __builtin_bswap64(w) : w;
}
// D of a bottom segment of `len` bases (at `start`): base o >= 1 of the segment has the boundary of the child's base pair
// (o - 1, o) — forward: the child's D at tstart + o; reversed: the pair is (tstart + len - o, tstart + len - 1 - o), whose boundary
// is D at tstart + len - o.  Base 0 begins the segment.  Sixteen lanes a segment, eight bases a lane; the lane at the segment's
// end takes the last eight bases (OR is idempotent).  Tracks are allocated eight bytes longer than their genome.
// (the body as a function of the thread's number and the number of threads: the kernel below, and the host-side check)
template <typename C>
HGX_DEV __forceinline__ void break_up_body(int64_t thread, int64_t threads, const BotRec<C> *__restrict__ bot, int64_t numBot, const BreakChildren &ch,
                                           int accumulate, uint8_t *__restrict__ D) {
    const int sub = (int)(thread & 15);
    const int64_t groupsTotal = threads >> 4;
    for (int64_t b = thread >> 4; b < numBot; b += groupsTotal) {
        const int64_t start = (int64_t)bot[b].start, len = (int64_t)bot[b + 1].start - start;
        for (int64_t o0 = 0; o0 < len; o0 += 128) {
            int64_t o = o0 + (int64_t)sub * 8;
            if (o >= len)
                continue;
            if (o + 8 > len && len >= 8)
                o = len - 8;
            const bool whole = o + 8 <= len;
            unsigned long long v = 0;
            if (accumulate) {
                if (whole)
                    __builtin_memcpy(&v, D + start + o, 8);
                else
                    for (int j = 0; j < 8 && o + j < len; ++j)
                        v |= (unsigned long long)D[start + o + j] << (8 * j);
            }
            for (int k = 0; k < ch.n; ++k) {
                const int32_t enc = ch.c[k].enc[b];
                if (enc < 0)
                    continue;
                const TopRec<C> *top = (const TopRec<C> *)ch.c[k].top;
                const int32_t t0 = enc >> 1;
                int32_t t = t0;
                do { // the slot's segment and its paralogy ring
                    const TopRec<C> tr = top[t];
                    const uint8_t *base = ch.c[k].track + (int64_t)tr.start;
                    const bool rev = (tr.parentEnc & 1) != 0;
                    if (whole) {
                        v |= break_word(rev ? base + len - o - 7 : base + o, rev);
                    } else {
                        for (int j = 0; j < 8 && o + j < len; ++j)
                            v |= (unsigned long long)base[rev ? len - o - j : o + j] << (8 * j);
                    }
                    t = ch.noRing ? -1 : tr.paralogy;
                } while (t >= 0 && t != t0);
            }
            if (o == 0)
                v |= 1ull;
            v &= 0x0101010101010101ull;
            if (whole) {
                __builtin_memcpy(D + start + o, &v, 8);
            } else {
                for (int j = 0; j < 8 && o + j < len; ++j)
                    D[start + o + j] = (uint8_t)(v >> (8 * j));
            }
        }
    }
}
template <typename C>
static __global__ void __launch_bounds__(256) k_break_up(const BotRec<C> *__restrict__ bot, int64_t numBot, BreakChildren ch, int accumulate,
                                                         uint8_t *__restrict__ D) {
    break_up_body<C>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, bot, numBot, ch, accumulate, D);
}
// F of a genome on the path below the top of the scope: a top segment with a parent takes the parent's F (read in the parent's
// orientation, as above); one without (an insertion) its own D; base 0 begins the segment.  pF: the parent's F; D: this genome's
// own D, or null when nothing in scope hangs under it.
template <typename C>
HGX_DEV __forceinline__ void break_down_body(int64_t thread, int64_t threads, const TopRec<C> *__restrict__ top, int64_t numTop,
                                             const BotRec<C> *__restrict__ pbot, const uint8_t *__restrict__ pF, const uint8_t *__restrict__ D,
                                             uint8_t *__restrict__ F, const int32_t *__restrict__ pEnc = nullptr) {
    const int sub = (int)(thread & 15);
    const int64_t groupsTotal = threads >> 4;
    for (int64_t t = thread >> 4; t < numTop; t += groupsTotal) {
        const TopRec<C> tr = top[t];
        const int64_t start = (int64_t)tr.start, len = (int64_t)top[t + 1].start - start;
        // (pEnc — --noDupes: only the segment the parent's slot names has a parent: k_sweep_down)
        const bool hasParent = tr.parentEnc >= 0 && (!pEnc || (int64_t)(pEnc[tr.parentEnc >> 1] >> 1) == t), rev = (tr.parentEnc & 1) != 0;
        const uint8_t *base = hasParent ? pF + (int64_t)pbot[tr.parentEnc >> 1].start : (D ? D + start : nullptr);
        for (int64_t o0 = 0; o0 < len; o0 += 128) {
            int64_t o = o0 + (int64_t)sub * 8;
            if (o >= len)
                continue;
            if (o + 8 > len && len >= 8)
                o = len - 8;
            const bool whole = o + 8 <= len;
            const bool r = hasParent && rev;
            unsigned long long v = 0;
            if (base) {
                if (whole) {
                    v = break_word(r ? base + len - o - 7 : base + o, r);
                } else {
                    for (int j = 0; j < 8 && o + j < len; ++j)
                        v |= (unsigned long long)base[r ? len - o - j : o + j] << (8 * j);
                }
            }
            if (o == 0)
                v |= 1ull;
            v &= 0x0101010101010101ull;
            if (whole) {
                __builtin_memcpy(F + start + o, &v, 8);
            } else {
                for (int j = 0; j < 8 && o + j < len; ++j)
                    F[start + o + j] = (uint8_t)(v >> (8 * j));
            }
        }
    }
}
template <typename C>
static __global__ void __launch_bounds__(256) k_break_down(const TopRec<C> *__restrict__ top, int64_t numTop, const BotRec<C> *__restrict__ pbot,
                                                           const uint8_t *__restrict__ pF, const uint8_t *__restrict__ D, uint8_t *__restrict__ F,
                                                           const int32_t *__restrict__ pEnc) {
    break_down_body<C>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, top, numTop, pbot, pF, D, F, pEnc);
}
// F of the genome at the top of the scope: its own D, and where its top segments begin (it has some when the scope ends below the root)
static __global__ void __launch_bounds__(256) k_break_top(const uint8_t *__restrict__ D, int64_t n, uint8_t *__restrict__ F) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        F[i] = D ? D[i] : 0;
}
template <typename C> static __global__ void __launch_bounds__(256) k_break_top_starts(const TopRec<C> *__restrict__ top, int64_t numTop, uint8_t *__restrict__ F) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < numTop; t += (int64_t)gridDim.x * blockDim.x)
        F[(int64_t)top[t].start] = 1;
}

// the coarse table MafSelect::locate starts from: out[b] = the segment that holds position b << shift (b < nb), out[nb] = the last segment
template <typename REC> static __global__ void __launch_bounds__(256) k_maf_locate_table(const REC *__restrict__ segs, int64_t nseg, int shift, uint32_t nb,
                                                                                         int32_t *__restrict__ out) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= nb; b += gridDim.x * blockDim.x) {
        if (b == nb) {
            out[b] = (int32_t)(nseg - 1);
            continue;
        }
        const int64_t pos = (int64_t)b << shift;
        int64_t lo = 0, hi = nseg; // the last segment that begins at or before pos
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)segs[mid].start <= pos)
                lo = mid;
            else
                hi = mid;
        }
        out[b] = (int32_t)lo;
    }
}

// ---- the marked columns of a chunk ----
// mark[c] = 1 when column c of the chunk may begin a run (its F, or the chunk's first column); rowsOf[c] = its rows (A) then, else 0
static __global__ void __launch_bounds__(256) k_maf_marks(const uint8_t *__restrict__ F, const int32_t *__restrict__ A, int32_t constRows, int64_t first,
                                                          uint32_t n, uint32_t *__restrict__ mark, uint32_t *__restrict__ rowsOf) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const bool m = c == 0 || F[first + c] != 0;
        mark[c] = m ? 1u : 0u;
        rowsOf[c] = m ? (uint32_t)(A ? A[first + c] : constRows) : 0u;
    }
}
// the marked columns in order: their column, the offset of their rows (candRow[nCand] = all rows: written by the caller's scan)
static __global__ void __launch_bounds__(256) k_maf_list(const uint32_t *__restrict__ mark, const uint32_t *__restrict__ markIdx,
                                                         const uint32_t *__restrict__ rowOff, uint32_t n, uint32_t *__restrict__ candCol,
                                                         uint32_t *__restrict__ candRow) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x)
        if (mark[c]) {
            candCol[markIdx[c]] = c;
            candRow[markIdx[c]] = rowOff[c];
        }
}

// ---- the rows of the marked columns: row k of a column by rank ----
struct MafRowParams {
    ColumnParams P;             // desc, ref, first (genome coordinate of the chunk's first column), noAncestors, the masks, error
    const int32_t *const *S;    // per genome: the reported bases in the tree below each base (own base included), or null: a constant
    const uint32_t *candCol;    // the marked columns (index in the chunk)
    const uint32_t *candRow;    // [nCand + 1] offsets of their rows
    uint32_t nCand;
    // coarse position -> segment table of the reference's tiling (k_maf_locate_table), or null: refLocate[b] = the segment that holds
    // position b << refLocateShift, one entry more for the last segment
    const int32_t *refLocate;
    int32_t refLocateShift;
};
static constexpr int MAF_LPC_LOG = 3; // lanes per marked column (config 3: 6.6 rows a column); a lane takes rows k, k + 8, ...

template <typename C> struct MafSelect {
    const ColumnParams &P;
    const int32_t *const *S;
    const int32_t *refLocate;
    int32_t refLocateShift;
    HGX_DEV MafSelect(const MafRowParams &p) : P(p.P), S(p.S), refLocate(p.refLocate), refLocateShift(p.refLocateShift) {
    }
    HGX_DEV __forceinline__ const TopRec<C> *top(int g) const {
        return (const TopRec<C> *)P.desc[g].top;
    }
    HGX_DEV __forceinline__ const BotRec<C> *bot(int g) const {
        return (const BotRec<C> *)P.desc[g].bot;
    }
    HGX_DEV __forceinline__ int32_t reported(int g) const { // colMapInsert's filters (halColumnIterator.cpp:802-812)
        return (!P.noAncestors || P.desc[g].numChildren == 0) && bit(P.targetMask, g) ? 1 : 0;
    }
    HGX_DEV __forceinline__ int32_t sizeAt(int g, int64_t pos) const {
        const int32_t *s = S[g];
        return s ? s[pos] : reported(g);
    }
    template <typename REC> HGX_DEV __forceinline__ int64_t posOf(const REC *segs, int32_t idx, int32_t so, bool rev) const {
        return !rev ? (int64_t)segs[idx].start + so : (int64_t)segs[idx + 1].start - 1 - so;
    }
    HGX_DEV __forceinline__ void emit(ColumnRow *dst, int g, int64_t pos, bool rev) const {
        RowVisitor v;
        v.dst = dst;
        v.desc = P.desc;
        v(g, pos, rev);
    }
    // index of the reference segment (top tiling, or bottom for a genome without one) holding position p
    HGX_DEV __forceinline__ int32_t locate(int64_t p) const {
        const GenomeDesc &RD = P.desc[P.ref];
        int64_t lo = 0, hi = RD.numTop > 0 ? RD.numTop : RD.numBot;
        if (refLocate) { // (twenty dependent loads of the search over a million segments down to three or four)
            const int64_t b = p >> refLocateShift;
            lo = refLocate[b];
            hi = (int64_t)refLocate[b + 1] + 1;
        }
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            const int64_t st = RD.numTop > 0 ? (int64_t)top(P.ref)[mid].start : (int64_t)bot(P.ref)[mid].start;
            if (st <= p)
                lo = mid;
            else
                hi = mid;
        }
        return (int32_t)lo;
    }
    // Row l of what hangs under the bottom segment (g, j) at offset so (iteration order, strand rev) — the CHILD frames of
    // updateParseDown / recursiveUpdate's bottom branch in slot order, `skip` left out (the path's own slot) — and on down:
    // one branch per level.  A CHILD block is: the slot's segment's base, then every other member of its paralogy ring with what
    // hangs under it, then what hangs under the slot's segment itself (updateChild pushes the parse-down first and the ring on
    // top of it, :607-640; updateNextTopDup goes round the ring, each member's subtree before the next member, :642-681).
    // Returns false when l lies behind everything here (l is then what is left).
    HGX_DEV bool below(ColumnRow *dst, int g, int32_t j, int32_t so, bool rev, int skip, int64_t &l) const {
        for (;;) {
            const GenomeDesc &D = P.desc[g];
            bool down = false;
            for (int i = 0; i < D.numChildren && !down; ++i) {
                if (i == skip)
                    continue;
                const int32_t enc = D.child[i][j];
                const int cg = D.childGenome[i];
                if (enc < 0 || !bit(P.scopeMask, cg))
                    continue;
                const TopRec<C> *CT = top(cg);
                const int32_t t0 = enc >> 1;
                const TopRec<C> c0 = CT[t0];
                const bool rev0 = rev ^ ((enc & 1) != 0);
                const int64_t pos0 = posOf(CT, t0, so, rev0);
                const int32_t own = reported(cg);
                const int64_t s0 = sizeAt(cg, pos0);
                if (own) {
                    if (l == 0) {
                        emit(dst, cg, pos0, rev0);
                        return true;
                    }
                    --l;
                }
                int32_t t = -1;
                bool trev = false;
                if (ringMember(dst, cg, CT, c0, t0, so, rev0, l, t, trev))
                    return true;
                if (t < 0) { // not in the ring's members: under the slot's segment itself?
                    if (l < s0 - own) {
                        t = t0;
                        trev = rev0;
                    } else {
                        l -= s0 - own;
                        continue;
                    }
                }
                // parse down from top segment (cg, t) (updateParseDown, :711-744) and go on one level lower
                const int32_t bp = CT[t].botParse;
                if (bp < 0)
                    return false;
                const int64_t pos = posOf(CT, t, so, trev);
                const BotRec<C> *B = bot(cg);
                int32_t jj = bp;
                while ((int64_t)B[jj + 1].start <= pos)
                    ++jj;
                so = !trev ? (int32_t)(pos - (int64_t)B[jj].start) : (int32_t)((int64_t)B[jj + 1].start - 1 - pos);
                g = cg;
                j = jj;
                rev = trev;
                skip = -1;
                down = true;
            }
            if (!down)
                return false;
        }
    }
    // The members of the paralogy ring of top segment t0 (record c0) of genome g behind t0, in ring order, each with what hangs
    // under it (updateNextTopDup).  Row l is a member's own base: emitted, true.  Row l lies under a member: t / trev name it and l
    // is the row among what hangs under it.  Otherwise l is reduced by the members' sizes and t stays -1.
    HGX_DEV __forceinline__ bool ringMember(ColumnRow *dst, int g, const TopRec<C> *T, const TopRec<C> &c0, int32_t t0, int32_t so, bool rev0,
                                               int64_t &l, int32_t &t, bool &trev) const {
        if (P.noDupes)
            return false; // (--noDupes: a ring's other members are in no column)
        const int32_t own = reported(g);
        TopRec<C> cur = c0;
        bool crev = rev0;
        while (cur.paralogy >= 0 && cur.paralogy != t0) {
            const int32_t nxt = cur.paralogy;
            const TopRec<C> nr = T[nxt];
            const bool nrev = crev ^ ((nr.parentEnc & 1) != (cur.parentEnc & 1));
            const int64_t npos = posOf(T, nxt, so, nrev);
            const int64_t sz = sizeAt(g, npos);
            if (l < sz) {
                if (own) {
                    if (l == 0) {
                        emit(dst, g, npos, nrev);
                        return true;
                    }
                    --l;
                }
                t = nxt;
                trev = nrev;
                return false;
            }
            l -= sz;
            cur = nr;
            crev = nrev;
        }
        return false;
    }
    // what hangs under top segment (g, t) at offset so: parse down, then `below`
    HGX_DEV bool underTop(ColumnRow *dst, int g, int32_t t, int32_t so, bool rev, int64_t &l) const {
        const TopRec<C> *T = top(g);
        const int32_t bp = T[t].botParse;
        if (bp < 0)
            return false;
        const int64_t pos = posOf(T, t, so, rev);
        const BotRec<C> *B = bot(g);
        int32_t j = bp;
        while ((int64_t)B[j + 1].start <= pos)
            ++j;
        const int32_t so2 = !rev ? (int32_t)(pos - (int64_t)B[j].start) : (int32_t)((int64_t)B[j + 1].start - 1 - pos);
        return below(dst, g, j, so2, rev, -1, l);
    }
    // row r of the column of reference position p (reference segment seg), `total` rows in all; false: the sizes do not add up
    HGX_DEV bool row(ColumnRow *dst, int32_t seg, int64_t p, int64_t r, int64_t total) const {
        int g = P.ref;
        const GenomeDesc &RD = P.desc[g];
        int64_t anc = 0; // reported bases among the reference base and the ancestors so far
        if (RD.numTop <= 0) {
            // the root as reference (recursiveUpdate's bottom branch, :302-353): its base, then every child
            if (reported(g)) {
                if (r == 0) {
                    emit(dst, g, p, false);
                    return true;
                }
                --r;
            }
            const BotRec<C> *B = bot(g);
            return below(dst, g, seg, (int32_t)(p - (int64_t)B[seg].start), false, -1, r);
        }
        int32_t t = seg;
        int32_t so = (int32_t)(p - (int64_t)top(g)[t].start);
        bool rev = false;
        if (reported(g)) {
            if (r == 0) {
                emit(dst, g, p, false);
                return true;
            }
            anc = 1;
        }
        {
            // what hangs under the reference base comes last (its parse-down frame lies at the bottom of the stack, :252-300)
            const int64_t u0 = sizeAt(g, p);
            const int64_t start0 = total - u0 + anc;
            if (r >= start0) {
                int64_t l = r - start0;
                return underTop(dst, g, t, so, false, l);
            }
        }
        for (;;) {
            // updateParent (:556-605): the parent's base, then — after everything above it — the siblings under the same bottom
            // segment and the other members of this top segment's paralogy ring
            const TopRec<C> *T = top(g);
            const TopRec<C> tr = T[t];
            const GenomeDesc &D = P.desc[g];
            if (tr.parentEnc < 0 || D.parent < 0 || !bit(P.scopeMask, D.parent))
                return false;
            const int pg = D.parent;
            const GenomeDesc &PD = P.desc[pg];
            const int32_t b = tr.parentEnc >> 1;
            if (P.noDupes && (PD.child[D.slotInParent][b] >> 1) != t)
                return false; // (--noDupes: only the segment the parent's slot names goes up — the column has no row up there)
            const bool brev = rev ^ ((tr.parentEnc & 1) != 0);
            const BotRec<C> *B = bot(pg);
            const int64_t apos = posOf(B, b, so, brev);
            if (reported(pg)) {
                if (r == anc) {
                    emit(dst, pg, apos, brev);
                    return true;
                }
                ++anc;
            }
            const int64_t start = total - sizeAt(pg, apos) + anc; // rows in front of what hangs off the path at this level
            if (r >= start) {
                int64_t l = r - start;
                if (below(dst, pg, b, so, brev, D.slotInParent, l))
                    return true;
                int32_t mt = -1;
                bool mrev = false;
                if (ringMember(dst, g, T, tr, t, so, rev, l, mt, mrev))
                    return true;
                if (mt < 0)
                    return false;
                return underTop(dst, g, mt, so, mrev, l);
            }
            // updateParseUp (:683-709): the same base in the parent's top tiling
            if (PD.parent < 0)
                return false;
            const int32_t tp = B[b].topParse;
            if (tp < 0)
                return false;
            const TopRec<C> *PT = top(pg);
            int32_t j = tp;
            while ((int64_t)PT[j + 1].start <= apos)
                ++j;
            so = !brev ? (int32_t)(apos - (int64_t)PT[j].start) : (int32_t)((int64_t)PT[j + 1].start - 1 - apos);
            g = pg;
            t = j;
            rev = brev;
        }
    }
};

template <typename C> HGX_DEV __forceinline__ void maf_rows_body(const MafRowParams &M, uint32_t nCand, ColumnRow *__restrict__ rows) {
    constexpr int LPC = 1 << MAF_LPC_LOG;
    MafSelect<C> sel(M);
    const int sub = (int)(threadIdx.x & (LPC - 1));
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> MAF_LPC_LOG;
    bool bad = false;
    for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> MAF_LPC_LOG; k < (int64_t)nCand; k += groups) {
        const uint32_t a = M.candRow[k], n = M.candRow[k + 1] - a;
        if ((uint32_t)sub >= n)
            continue;
        const int64_t p = M.P.first + (int64_t)M.candCol[k];
        const int32_t seg = sel.locate(p);
        for (uint32_t r = (uint32_t)sub; r < n; r += LPC)
            if (!sel.row(rows + a + r, seg, p, (int64_t)r, (int64_t)n))
                bad = true;
    }
    if (bad)
        *M.P.error = 2;
}
template <typename C> static __global__ void __launch_bounds__(256) k_maf_rows(MafRowParams M, ColumnRow *__restrict__ rows) {
    maf_rows_body<C>(M, M.nCand, rows);
}

// ---- a chunk in one pass: no count on the host between the launches ----
// What the chunk's launches tell each other (and, at the chunk's end, the host) lies in device memory, cleared before the chunk.
struct MafChunkCtl {
    unsigned int ticket[4]; // the tiles' numbers of the chunk's one-pass scans (hgx_scan_kernels.hpp)
    unsigned int nCand, nHeads, error, nSeg; // error: 1 frame stack, 2 sizes, 3 --unique needs the walk (as P.error), MAF_ERR_* below; nSeg: --unique's stretches
    unsigned long long totalRows, totalHeadRows;
};
static constexpr unsigned MAF_ERR_TOO_MANY_ROWS = 4, MAF_ERR_ROWS_ROOM = 5, MAF_ERR_OUT_ROOM = 6;
static constexpr uint32_t MAF_MARK_TILE = 2048; // columns a tile of k_maf_mark_list (256 threads, eight columns each)

// k_maf_marks, both scans and k_maf_list in one launch: the marked columns of the chunk in order, the offsets of their rows, their
// number and their rows' number (ctl), and the chunk's head marks cleared.  grid = tiles of MAF_MARK_TILE columns.
static __global__ void __launch_bounds__(256) k_maf_mark_list(const uint8_t *__restrict__ F, const int32_t *__restrict__ A, int32_t constRows, int64_t first,
                                                              uint32_t n, MafChunkCtl *ctl, unsigned long long *tiles, uint32_t *__restrict__ candCol,
                                                              uint32_t *__restrict__ candRow, uint8_t *__restrict__ head) {
    const unsigned tile = lb_take_tile(&ctl->ticket[0]);
    const unsigned numTiles = (n + MAF_MARK_TILE - 1) / MAF_MARK_TILE;
    if (tile >= numTiles)
        return;
    const uint32_t c0 = tile * MAF_MARK_TILE + threadIdx.x * 8;
    uint32_t rowsOf[8];
    unsigned marks = 0;
    unsigned long long cnt = 0, w = 0;
    for (int j = 0; j < 8; ++j) {
        const uint32_t c = c0 + (uint32_t)j;
        rowsOf[j] = 0;
        if (c < n && (c == 0 || F[first + c] != 0)) {
            marks |= 1u << j;
            rowsOf[j] = (uint32_t)(A ? A[first + c] : constRows);
            ++cnt;
            w += rowsOf[j];
        }
        if (c < n)
            head[c] = 0;
    }
    const LbResult r = lb_scan_tile(tile, cnt, w, tiles);
    unsigned long long idx = r.exC, off = r.exW;
    for (int j = 0; j < 8; ++j)
        if (marks & (1u << j)) {
            candCol[idx] = c0 + (uint32_t)j;
            candRow[idx] = (uint32_t)off;
            ++idx;
            off += rowsOf[j];
        }
    if (tile == numTiles - 1 && threadIdx.x == 0) { // (the last tile's sums are the chunk's)
        const unsigned long long nc = r.baseC + r.tileC, nr = r.baseW + r.tileW;
        ctl->nCand = (unsigned int)nc;
        ctl->totalRows = nr;
        candRow[nc] = (uint32_t)nr;
        if (nr >= (1ull << 32))
            ctl->error = MAF_ERR_TOO_MANY_ROWS;
    }
}
// k_maf_rows with the marked columns' number taken from ctl; rowsRoom: the rows `rows` holds
template <typename C> static __global__ void __launch_bounds__(256) k_maf_rows_ctl(MafRowParams M, MafChunkCtl *ctl, unsigned long long rowsRoom,
                                                                                  ColumnRow *__restrict__ rows) {
    if (ctl->error >= MAF_ERR_TOO_MANY_ROWS)
        return;
    if (ctl->totalRows > rowsRoom) { // (every thread sees the same two numbers: nobody writes)
        if (blockIdx.x == 0 && threadIdx.x == 0)
            ctl->error = MAF_ERR_ROWS_ROOM;
        return;
    }
    M.P.error = &ctl->error;
    maf_rows_body<C>(M, ctl->nCand, rows);
}

// What the walk wants of a row (RunMachine::PRow, hgx_columns_host.cpp; MafRenderRow): the key of its base in its sequence on its
// strand, the rank of its sequence — initEntry with a base (halMafBlock.cpp:84-112) and SequenceLess (halColumnIterator.h:45-50),
// which the host's threads used to work out row by row (describe).  rankBase[g]: the rank of genome g's first sequence (a genome's
// sequences have consecutive ranks).
HGX_DEV __forceinline__ void maf_describe(const GenomeDesc *__restrict__ desc, const int32_t *__restrict__ rankBase, const ColumnRow &r, int64_t &key,
                                          int32_t &rank) {
    const GenomeDesc &G = desc[r.genome];
    int32_t s = 0;
    if (G.numSeq > 1) { // the sequence that holds the base (Genome::getSequenceBySite)
        int32_t lo = 0, hi = G.numSeq;
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if (G.seqStart[mid] <= r.pos)
                lo = mid;
            else
                hi = mid;
        }
        s = lo;
    }
    const int64_t at = r.pos - G.seqStart[s], len = G.seqStart[s + 1] - G.seqStart[s];
    key = r.rev ? ((len - 1 - at) << 1) | 1 : at << 1;
    rank = rankBase[r.genome] + s;
}
// what a wavefront's lanes wrote to LDS is read by its other lanes behind this point
__device__ __forceinline__ void maf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
struct MafHeadRow { // = RunMachine::PRow
    int64_t key;
    int32_t rank;
    uint32_t ord;
};
// k_maf_heads, both scans and k_maf_gather in one launch, and the rows as the walk wants them: which marked columns begin a run
// (compared with the marked column before them), the heads' marks, columns and row offsets, their rows described and sorted the way
// the column map holds them (by sequence, a sequence's bases in the walk's order).  head: device memory; headOff, headCol, out: where
// k_maf_ship takes them to the host.  Tiles of 256 marked columns, taken by ticket.
static constexpr int MAF_SORT_CACHE = 32; // a column's ranks kept in LDS while its rows are placed (longer columns: worked out again)
static __global__ void __launch_bounds__(256) k_maf_heads_out(const uint32_t *__restrict__ candCol, const uint32_t *__restrict__ candRow,
                                                              const ColumnRow *__restrict__ rows, MafChunkCtl *ctl, unsigned long long *tiles,
                                                              const GenomeDesc *__restrict__ desc, const int32_t *__restrict__ rankBase, uint32_t headRoom,
                                                              unsigned long long outRoom, uint8_t *__restrict__ head, uint32_t *__restrict__ headOff,
                                                              uint32_t *__restrict__ headCol, MafHeadRow *__restrict__ out) {
    __shared__ uint32_t sA[256], sN[256], sO[256];
    __shared__ int32_t sRank[32][MAF_SORT_CACHE];
  for (;;) { // (a workgroup takes tiles until there are none left: the grid need not know how many marked columns the chunk has)
    unsigned failed;
    const unsigned tile = lb_take_tile(&ctl->ticket[1], &ctl->error, failed);
    if (failed)
        return;
    const uint32_t nCand = ctl->nCand;
    const unsigned numTiles = nCand ? (nCand + 255) / 256 : 1;
    if (tile >= numTiles)
        return;
    const uint32_t k = tile * 256 + threadIdx.x;
    uint32_t a = 0, n = 0;
    bool isHead = false;
    if (k < nCand) {
        a = candRow[k];
        n = candRow[k + 1] - a;
        isHead = k == 0;
        if (!isHead) { // the marked column before, advanced by the distance (k_maf_heads)
            const uint32_t pa = candRow[k - 1];
            const int64_t d = (int64_t)candCol[k] - (int64_t)candCol[k - 1];
            isHead = a - pa != n;
            for (uint32_t i = 0; i < n && !isHead; ++i) {
                const ColumnRow r = rows[a + i], q = rows[pa + i];
                isHead = r.genome != q.genome || r.rev != q.rev || r.pos != (q.rev ? q.pos - d : q.pos + d);
            }
        }
    }
    const LbResult r = lb_scan_tile(tile, isHead ? 1u : 0u, isHead ? n : 0u, tiles);
    const bool room = r.baseC + r.tileC <= headRoom && r.baseW + r.tileW <= outRoom && r.baseW + r.tileW < (1ull << 32);
    if (tile == numTiles - 1 && threadIdx.x == 0) {
        ctl->nHeads = (unsigned int)(r.baseC + r.tileC);
        ctl->totalHeadRows = r.baseW + r.tileW;
        if (!room)
            ctl->error = MAF_ERR_OUT_ROOM;
    }
    if (!room) { // (the tiles behind this one find no room either; the ones in front have written what nobody will read)
        if (threadIdx.x == 0)
            ctl->error = MAF_ERR_OUT_ROOM;
        return;
    }
    sA[threadIdx.x] = a;
    sN[threadIdx.x] = isHead ? n : 0;
    sO[threadIdx.x] = (uint32_t)r.exW;
    if (isHead) {
        head[candCol[k]] = 1;
        headOff[r.exC] = (uint32_t)r.exW;
        headCol[r.exC] = candCol[k];
    }
    __syncthreads();
    // the rows: eight lanes a head, a lane a row at a time — its place among the column's rows is the number of rows that sort in
    // front of it (by rank, then by the walk's order)
    const int sub = (int)(threadIdx.x & 7), grp = (int)(threadIdx.x >> 3);
    for (int c = grp; c < 256; c += 32) {
        const uint32_t hn = sN[c];
        if (hn == 0)
            continue; // (the group's lanes agree)
        const uint32_t ha = sA[c], ho = sO[c];
        const bool cached = hn <= (uint32_t)MAF_SORT_CACHE;
        if (cached)
            for (uint32_t i = (uint32_t)sub; i < hn; i += 8) {
                int64_t key;
                int32_t rank;
                maf_describe(desc, rankBase, rows[ha + i], key, rank);
                sRank[grp][i] = rank;
            }
        // (the eight lanes of a group are lanes of one wavefront: what they wrote to LDS is there when they read it behind this point
        // only after a barrier of the wavefront — the groups of a workgroup walk the same number of heads apart from the last round, so a
        // workgroup barrier would not do; the wavefront's own is enough)
        maf_wave_sync();
        for (uint32_t i = (uint32_t)sub; i < hn; i += 8) {
            int64_t key;
            int32_t rank;
            maf_describe(desc, rankBase, rows[ha + i], key, rank);
            uint32_t place = 0;
            for (uint32_t j = 0; j < hn; ++j) {
                int32_t rj;
                if (cached) {
                    rj = sRank[grp][j];
                } else {
                    int64_t kj;
                    maf_describe(desc, rankBase, rows[ha + j], kj, rj);
                }
                place += (rj < rank || (rj == rank && j < i)) ? 1u : 0u;
            }
            out[ho + place] = MafHeadRow{key, rank, i};
        }
        maf_wave_sync();
    }
  }
}

// What k_maf_heads_out left in HBM goes to the host's page-locked memory from here, by the counts in ctl: consecutive lanes write
// consecutive sixteen bytes — a wavefront's store is a kilobyte the link takes in one piece (the heads' rows written to the host
// row by row, each at its sorted place, took three times the launch this and the one above take together).
static __global__ void __launch_bounds__(256) k_maf_ship(const MafChunkCtl *ctl, const uint32_t *__restrict__ headOff, const uint32_t *__restrict__ headCol,
                                                         const MafHeadRow *__restrict__ out, uint32_t *__restrict__ hostHeadOff,
                                                         uint32_t *__restrict__ hostHeadCol, MafHeadRow *__restrict__ hostOut) {
    if (ctl->error)
        return;
    const uint32_t nHeads = ctl->nHeads;
    const unsigned long long rows = ctl->totalHeadRows;
    const unsigned long long threads = (unsigned long long)gridDim.x * blockDim.x, me = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long i = me; i < rows; i += threads)
        hostOut[i] = out[i];
    for (unsigned long long i = me; i < nHeads; i += threads) {
        hostHeadOff[i] = headOff[i];
        hostHeadCol[i] = headCol[i];
    }
}

// ---- which marked columns are heads ----
// marked column k continues marked column k - 1 (no boundary lies between the two, so the columns in between are k - 1's
// advanced) when it has the same rows, each d = candCol[k] - candCol[k - 1] bases further on its strand
static __global__ void __launch_bounds__(256) k_maf_heads(const uint32_t *__restrict__ candCol, const uint32_t *__restrict__ candRow,
                                                          const ColumnRow *__restrict__ rows, uint32_t nCand, uint32_t *__restrict__ isHead,
                                                          uint32_t *__restrict__ headRows) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nCand; k += gridDim.x * blockDim.x) {
        const uint32_t a = candRow[k], n = candRow[k + 1] - a;
        bool head = k == 0;
        if (!head) {
            const uint32_t pa = candRow[k - 1];
            const int64_t d = (int64_t)candCol[k] - (int64_t)candCol[k - 1];
            head = a - pa != n;
            for (uint32_t i = 0; i < n && !head; ++i) {
                const ColumnRow r = rows[a + i], q = rows[pa + i];
                head = r.genome != q.genome || r.rev != q.rev || r.pos != (q.rev ? q.pos - d : q.pos + d);
            }
        }
        isHead[k] = head ? 1u : 0u;
        headRows[k] = head ? n : 0u;
    }
}
// the heads' rows packed, their offsets, and the per-column marks of the chunk (cleared by the caller)
static __global__ void __launch_bounds__(256) k_maf_gather(const uint32_t *__restrict__ candCol, const uint32_t *__restrict__ candRow,
                                                           const ColumnRow *__restrict__ rows, uint32_t nCand, const uint32_t *__restrict__ isHead,
                                                           const uint32_t *__restrict__ headIdx, const uint32_t *__restrict__ headRowOff,
                                                           uint8_t *__restrict__ head, uint32_t *__restrict__ headOffset, ColumnRow *__restrict__ out) {
    constexpr int LPC = 1 << MAF_LPC_LOG;
    const int sub = (int)(threadIdx.x & (LPC - 1));
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> MAF_LPC_LOG;
    for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> MAF_LPC_LOG; k < (int64_t)nCand; k += groups) {
        if (!isHead[k])
            continue;
        const uint32_t a = candRow[k], n = candRow[k + 1] - a, o = headRowOff[k];
        if (sub == 0) {
            head[candCol[k]] = 1;
            headOffset[headIdx[k]] = o;
        }
        for (uint32_t i = (uint32_t)sub; i < n; i += LPC)
            out[o + i] = rows[a + i];
    }
}

// ---- hal2maf --unique from the marked columns' rows ----
// Which columns the iterator's visit cache lets it walk, and which of them hal2maf writes, is a function of the column's REFERENCE
// bases (hgx_column_kernels.hpp: k_column_unique_count has the derivation): with f the range's first column and R(p) the reference
// bases of column p's walk, p is passed over when a base of R(p) lies in [f, p), is walked without being written when none does but
// one lies left of f, and is written otherwise.  Inside a run the rows only move — forward rows with the column, reverse rows
// against it — so the class of column h + j of the run that begins at marked column h changes where a reverse copy of the reference
// base crosses it or a copy crosses f: a run falls into stretches of one class, found from the marked column's rows alone.
struct UniqueSeg {
    uint32_t col;  // first column of the stretch (in the chunk)
    uint32_t len;  // its columns
    uint32_t cand; // the marked column whose run it lies in
    uint32_t j;    // col - that marked column
    uint32_t cls;  // COL_SKIPPED / COL_WRITTEN / COL_KEYS_ONLY
};
static constexpr int UNIQUE_MAX_REF_ROWS = 32;
struct UniqueParams {
    const uint32_t *candCol, *candRow;
    const ColumnRow *rows;
    uint32_t nCand, n; // marked columns, columns of the chunk
    int64_t first, f;  // genome coordinate of the chunk's first column, of the range's first column
    int32_t ref;
    int32_t maxRefRows; // <= UNIQUE_MAX_REF_ROWS (the tests lower it: HGX_MAF_UNIQUE_MAX_REF)
    // 1: a stretch that is walked for its keys ships its FIRST column's rows only and marks the columns behind it as passed over.  What
    // such a column does to the walk — its sequences become keys of the column map and count as seen in the block being made
    // (RunMachine::walkChunk) — is the same for every column of the stretch: inside a run the rows only move, and a run ends where a
    // sequence does (a segment boundary).  A slice deep inside a genome has such a column for every copy left of it: a fifth of
    // hgx_maf_export_multi's columns, each with all its rows.  0: every column's rows, as the column walk ships them (the chunk
    // that is held against the walk).
    int32_t collapseKeysOnly;
    unsigned int *error; // 3: a column with more reference bases than UNIQUE_MAX_REF_ROWS, or none (the reference is not reported)
};
// emit(j, len, cls) for every stretch of marked column k's run; false: the column's reference rows cannot be held
template <typename F> HGX_DEV __forceinline__ bool unique_stretches(const UniqueParams &U, uint32_t k, F emit) {
    const uint32_t a = U.candRow[k], nr = U.candRow[k + 1] - a;
    const int64_t h = (int64_t)U.candCol[k], L = (int64_t)(k + 1 < U.nCand ? U.candCol[k + 1] : U.n) - h;
    const int64_t p0 = U.first + h;
    int64_t pos[UNIQUE_MAX_REF_ROWS];
    bool rev[UNIQUE_MAX_REF_ROWS];
    int m = 0;
    bool self = false;
    for (uint32_t i = 0; i < nr; ++i) {
        const ColumnRow r = U.rows[a + i];
        if (r.genome != U.ref)
            continue;
        if (r.pos == p0 && !r.rev) {
            self = true; // (the column's own base: never left of itself)
            continue;
        }
        if (m >= U.maxRefRows)
            return false;
        pos[m] = r.pos;
        rev[m] = r.rev != 0;
        ++m;
    }
    if (!self)
        return false; // (the reference's bases are not among the rows: a filter keeps them out)
    int64_t j = 0;
    int64_t start = 0;
    uint32_t cur = 0;
    bool have = false;
    while (j < L) {
        // the class of column h + j, and the first column behind it where a copy crosses the column or the range's beginning
        const int64_t p = p0 + j;
        bool inRange = false, left = false;
        int64_t next = L;
        for (int i = 0; i < m; ++i) {
            const int64_t x = rev[i] ? pos[i] - j : pos[i] + j;
            if (x < p) {
                if (x >= U.f)
                    inRange = true;
                else
                    left = true;
            }
            int64_t b;
            if (!rev[i]) {
                b = U.f - pos[i]; // from here on the copy lies at or behind f
                if (b > j && b < next)
                    next = b;
            } else {
                b = (pos[i] - p0) / 2 + 1; // from here on the copy lies left of the column (pos - j < p0 + j)
                if (pos[i] >= p0 && b > j && b < next)
                    next = b;
                b = pos[i] - U.f + 1; // from here on the copy lies left of f
                if (b > j && b < next)
                    next = b;
            }
        }
        const uint32_t cls = inRange ? COL_SKIPPED : left ? COL_KEYS_ONLY : COL_WRITTEN;
        if (have && cls != cur) {
            emit(start, j - start, cur);
            start = j;
        }
        cur = cls;
        have = true;
        j = next;
    }
    if (have)
        emit(start, L - start, cur);
    return true;
}
static __global__ void __launch_bounds__(256) k_unique_count(UniqueParams U, uint32_t *__restrict__ segCount) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < U.nCand; k += gridDim.x * blockDim.x) {
        uint32_t n = 0;
        if (!unique_stretches(U, k, [&](int64_t, int64_t, uint32_t) { ++n; }))
            *U.error = 3;
        segCount[k] = n;
    }
}
static __global__ void __launch_bounds__(256) k_unique_stretches(UniqueParams U, const uint32_t *__restrict__ segOff, UniqueSeg *__restrict__ seg) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < U.nCand; k += gridDim.x * blockDim.x) {
        uint32_t at = segOff[k];
        unique_stretches(U, k, [&](int64_t j, int64_t len, uint32_t cls) {
            seg[at++] = UniqueSeg{(uint32_t)(U.candCol[k] + j), (uint32_t)len, k, (uint32_t)j, cls};
        });
    }
}
// a stretch's share of the output: a written stretch that begins a run of written columns ships its first column's rows (a head); a
// stretch walked without being written ships every column's rows (their sequences become keys of the column map, column by column,
// as the walk delivers them); units = entries of headOffset
static __global__ void __launch_bounds__(256) k_unique_units(UniqueParams U, const UniqueSeg *__restrict__ seg, uint32_t nSeg, uint32_t *__restrict__ units,
                                                             uint32_t *__restrict__ unitRows) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nSeg; s += gridDim.x * blockDim.x) {
        const UniqueSeg g = seg[s];
        const uint32_t a = U.candRow[g.cand], nr = U.candRow[g.cand + 1] - a;
        uint32_t u = 0;
        if (g.cls == COL_KEYS_ONLY) {
            u = U.collapseKeysOnly ? 1u : g.len;
        } else if (g.cls == COL_WRITTEN) {
            bool head = g.col == 0 || g.j > 0 || s == 0 || seg[s - 1].cls != COL_WRITTEN;
            if (!head) { // the marked column before, advanced by the distance (k_maf_heads' test)
                const uint32_t k = g.cand, pa = U.candRow[k - 1];
                const int64_t d = (int64_t)U.candCol[k] - (int64_t)U.candCol[k - 1];
                head = a - pa != nr;
                for (uint32_t i = 0; i < nr && !head; ++i) {
                    const ColumnRow r = U.rows[a + i], q = U.rows[pa + i];
                    head = r.genome != q.genome || r.rev != q.rev || r.pos != (q.rev ? q.pos - d : q.pos + d);
                }
            }
            u = head ? 1 : 0;
        }
        units[s] = u;
        unitRows[s] = u * nr;
    }
}
// marks, offsets and rows of the chunk from the stretches (head: cleared by the caller)
static __global__ void __launch_bounds__(256) k_unique_gather(UniqueParams U, const GenomeDesc *__restrict__ desc, const UniqueSeg *__restrict__ seg, uint32_t nSeg,
                                                              const uint32_t *__restrict__ units, const uint32_t *__restrict__ unitOff,
                                                              const uint32_t *__restrict__ rowOff, uint8_t *__restrict__ head,
                                                              uint32_t *__restrict__ headOffset, ColumnRow *__restrict__ out) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nSeg; s += gridDim.x * blockDim.x) {
        const UniqueSeg g = seg[s];
        const uint32_t a = U.candRow[g.cand], nr = U.candRow[g.cand + 1] - a;
        if (g.cls == COL_SKIPPED) {
            for (uint32_t t = 0; t < g.len; ++t)
                head[g.col + t] = 2;
            continue;
        }
        if (g.cls == COL_KEYS_ONLY)
            for (uint32_t t = 0; t < g.len; ++t)
                head[g.col + t] = U.collapseKeysOnly && t > 0 ? 2 : 3;
        const uint32_t u = units[s];
        if (g.cls == COL_WRITTEN && u)
            head[g.col] = 1;
        for (uint32_t t = 0; t < u; ++t) {
            const uint32_t o = rowOff[s] + t * nr;
            headOffset[unitOff[s] + t] = o;
            const int64_t shift = (int64_t)g.j + t;
            for (uint32_t i = 0; i < nr; ++i) {
                const ColumnRow r = U.rows[a + i];
                if (shift == 0) {
                    out[o + i] = r;
                } else { // the same base `shift` columns on along its strand, read anew
                    RowVisitor v;
                    v.dst = out + o + i;
                    v.desc = desc;
                    v(r.genome, r.rev ? r.pos - shift : r.pos + shift, r.rev != 0);
                }
            }
        }
    }
}


// ---- --unique in the stream: the stretches listed and the units shipped in one launch each ----
// k_unique_count, the scan and k_unique_stretches: a lane a marked column counts its run's stretches, the tiles' counts are scanned in
// the same launch, the lane goes through its run again and writes them.  (U.nCand, U.error: taken from ctl here.)
static __global__ void __launch_bounds__(256) k_unique_stretch_list(UniqueParams U, MafChunkCtl *ctl, unsigned long long *tiles, UniqueSeg *__restrict__ seg,
                                                                    uint32_t segRoom) {
    for (;;) {
        unsigned failed;
        const unsigned tile = lb_take_tile(&ctl->ticket[2], &ctl->error, failed);
        if (failed)
            return;
        U.nCand = ctl->nCand;
        U.error = &ctl->error;
        const unsigned numTiles = U.nCand ? (U.nCand + 255) / 256 : 1;
        if (tile >= numTiles)
            return;
        const uint32_t k = tile * 256 + threadIdx.x;
        uint32_t cnt = 0;
        bool bad = false;
        if (k < U.nCand)
            bad = !unique_stretches(U, k, [&](int64_t, int64_t, uint32_t) { ++cnt; });
        if (bad)
            cnt = 0;
        const LbResult r = lb_scan_tile(tile, cnt, 0, tiles);
        const bool room = r.baseC + r.tileC <= segRoom;
        if (bad)
            ctl->error = 3; // (a column whose reference rows a lane cannot hold: this batch is the walk's)
        if (tile == numTiles - 1 && threadIdx.x == 0)
            ctl->nSeg = (unsigned int)(r.baseC + r.tileC);
        if (!room) {
            if (threadIdx.x == 0)
                ctl->error = MAF_ERR_OUT_ROOM;
            return;
        }
        if (k < U.nCand && !bad) {
            uint32_t at = (uint32_t)r.exC;
            unique_stretches(U, k, [&](int64_t j, int64_t len, uint32_t cls) {
                seg[at++] = UniqueSeg{(uint32_t)(U.candCol[k] + j), (uint32_t)len, k, (uint32_t)j, cls};
            });
        }
    }
}
// k_unique_units, both scans and k_unique_gather, the rows as the walk wants them (k_maf_heads_out's form): a lane a stretch — its
// units (a written stretch that begins a run of written columns: its first column; a stretch walked for its keys: every column, or
// the first one only: UniqueParams::collapseKeysOnly), the marks of its columns, its units' columns, row offsets and rows, every row
// moved along its strand to the unit's column, described and sorted.
static constexpr int UNIQUE_SORT_ROWS = 24; // a unit's ranks kept by its lane (longer columns: worked out again)
static __global__ void __launch_bounds__(256) k_unique_out(UniqueParams U, MafChunkCtl *ctl, unsigned long long *tiles, const UniqueSeg *__restrict__ seg,
                                                           const GenomeDesc *__restrict__ desc, const int32_t *__restrict__ rankBase, uint32_t headRoom,
                                                           unsigned long long outRoom, uint8_t *__restrict__ head, uint32_t *__restrict__ headOff,
                                                           uint32_t *__restrict__ headCol, MafHeadRow *__restrict__ out) {
    for (;;) {
        unsigned failed;
        const unsigned tile = lb_take_tile(&ctl->ticket[3], &ctl->error, failed);
        if (failed)
            return;
        const uint32_t nSeg = ctl->nSeg;
        const unsigned numTiles = nSeg ? (nSeg + 255) / 256 : 1;
        if (tile >= numTiles)
            return;
        const uint32_t s = tile * 256 + threadIdx.x;
        UniqueSeg g{0, 0, 0, 0, COL_SKIPPED};
        uint32_t a = 0, nr = 0, u = 0;
        if (s < nSeg) {
            g = seg[s];
            a = U.candRow[g.cand];
            nr = U.candRow[g.cand + 1] - a;
            if (g.cls == COL_KEYS_ONLY) {
                u = U.collapseKeysOnly ? 1u : g.len;
            } else if (g.cls == COL_WRITTEN) {
                bool isHead = g.col == 0 || g.j > 0 || s == 0 || seg[s - 1].cls != COL_WRITTEN;
                if (!isHead) { // the marked column before, advanced by the distance (k_maf_heads' test)
                    const uint32_t k = g.cand, pa = U.candRow[k - 1];
                    const int64_t d = (int64_t)U.candCol[k] - (int64_t)U.candCol[k - 1];
                    isHead = a - pa != nr;
                    for (uint32_t i = 0; i < nr && !isHead; ++i) {
                        const ColumnRow r = U.rows[a + i], q = U.rows[pa + i];
                        isHead = r.genome != q.genome || r.rev != q.rev || r.pos != (q.rev ? q.pos - d : q.pos + d);
                    }
                }
                u = isHead ? 1 : 0;
            }
        }
        const LbResult r = lb_scan_tile(tile, u, (unsigned long long)u * nr, tiles);
        const bool room = r.baseC + r.tileC <= headRoom && r.baseW + r.tileW <= outRoom && r.baseW + r.tileW < (1ull << 32);
        if (tile == numTiles - 1 && threadIdx.x == 0) {
            ctl->nHeads = (unsigned int)(r.baseC + r.tileC);
            ctl->totalHeadRows = r.baseW + r.tileW;
        }
        if (!room) {
            if (threadIdx.x == 0)
                ctl->error = MAF_ERR_OUT_ROOM;
            return;
        }
        if (s >= nSeg)
            continue;
        if (g.cls == COL_SKIPPED) {
            for (uint32_t t = 0; t < g.len; ++t)
                head[g.col + t] = 2;
            continue;
        }
        if (g.cls == COL_KEYS_ONLY)
            for (uint32_t t = 0; t < g.len; ++t)
                head[g.col + t] = U.collapseKeysOnly && t > 0 ? 2 : 3;
        if (g.cls == COL_WRITTEN && u)
            head[g.col] = 1;
        for (uint32_t t = 0; t < u; ++t) {
            const uint32_t unit = (uint32_t)r.exC + t, o = (uint32_t)r.exW + t * nr;
            headOff[unit] = o;
            headCol[unit] = g.col + t;
            const int64_t shift = (int64_t)g.j + t; // the same bases `shift` columns on along their strands
            int32_t rk[UNIQUE_SORT_ROWS];
            const bool cached = nr <= (uint32_t)UNIQUE_SORT_ROWS;
            auto moved = [&](uint32_t i) {
                ColumnRow x = U.rows[a + i];
                x.pos = x.rev ? x.pos - shift : x.pos + shift;
                return x;
            };
            if (cached)
                for (uint32_t i = 0; i < nr; ++i) {
                    int64_t key;
                    maf_describe(desc, rankBase, moved(i), key, rk[i]);
                }
            for (uint32_t i = 0; i < nr; ++i) {
                int64_t key;
                int32_t rank;
                maf_describe(desc, rankBase, moved(i), key, rank);
                uint32_t place = 0;
                for (uint32_t j = 0; j < nr; ++j) {
                    int32_t rj;
                    if (cached) {
                        rj = rk[j];
                    } else {
                        int64_t kj;
                        maf_describe(desc, rankBase, moved(j), kj, rj);
                    }
                    place += (rj < rank || (rj == rank && j < i)) ? 1u : 0u;
                }
                out[o + place] = MafHeadRow{key, rank, i};
            }
        }
    }
}

} // namespace hgx
