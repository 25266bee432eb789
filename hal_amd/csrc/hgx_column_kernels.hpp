// Column engine kernels (gfx950): the per-base closure of the reference's ColumnIterator
// (api/impl/halColumnIterator.cpp, maxInsertLength == 0 and unique == false, where columns are independent,
// :785-787).  The reference redoes a recursive walk with heap-allocated linked iterators for every base; here
// one lane owns one reference base and runs the same depth-first walk with an explicit frame stack.  The 64
// lanes of a wavefront hold 64 consecutive bases, which sit in the same segments almost everywhere, so their
// table reads collapse into broadcast loads and the lanes stay converged; they diverge only at segment
// boundaries.  The walk order is the reference's (updateParent, then paralogs, then the parse-down subtree;
// :246-355, :556-744) so that rows come out in ColumnMap insertion order for the MAF writer.
#pragma once
#include <type_traits>
#include "hgx_scan_kernels.hpp"
#include "hgx_device.hpp"
#include <hip/hip_runtime.h>

// (HGX_DEV: the walk and the row selection are plain functions of the tables; tests/cpp/maf_select_check.cpp compiles them for
// the host — -DHGX_DEV="__host__ __device__" — and runs them there against each other)
#ifndef HGX_DEV
#define HGX_DEV __device__
#endif

namespace hgx {

struct ColumnParams {
    const GenomeDesc *desc;
    int32_t numGenomes;
    int32_t ref;
    int64_t first; // genome coordinate of the first column
    int64_t count; // number of columns
    int64_t step;  // distance between consecutive columns (halAlignmentDepth --step)
    int32_t noDupes, noAncestors, onlyOrthologs;
    int32_t noGapEvents; // hgx_gap_kernels.hpp: the walk reports no deletions / insertions (maxInsertLength == 0)
    const unsigned long long *scopeMask;  // device, ceil(numGenomes / 64) words: genomes the walk may enter (all ones when no targets)
    const unsigned long long *targetMask; // device: genomes whose bases are reported
    unsigned int *error;              // set to 1 on frame-stack overflow
    unsigned long long *derefs;       // optional {top, bottom} segment records logically dereferenced (roofline accounting)
};

// a visitor that also wants every base the walk reaches, whatever the noAncestors / targets filters say of its genome
// (UniqueVisitor below): specialise to true and give it  void raw(int genome, int64_t pos)
template <typename V> struct VisitorWantsRaw {
    static constexpr bool value = false;
};

static constexpr int COL_STACK = 64; // frames per lane (scratch; an LDS-resident lower part was measured slower: occupancy)

enum : uint32_t { FR_UP = 0, FR_PARSEUP = 1, FR_CHILD = 2, FR_RING = 3, FR_PARSEDOWN = 4 };

struct Frame {
    int32_t idx;   // segment index (top for UP/RING/PARSEDOWN, bottom for PARSEUP/CHILD)
    int32_t so;    // offset of the base inside the segment, in iteration order
    int32_t extra; // CHILD: slot; RING: index of the ring's first member
    uint32_t meta; // kind | rev << 3 | genome << 4
};

HGX_DEV __forceinline__ bool bit(const unsigned long long *m, int g) {
    return (m[g >> 6] >> (g & 63)) & 1ull;
}

template <typename C, bool STATS = false> struct ColumnWalker {
    const ColumnParams &P;
    Frame stack[COL_STACK];
    int sp = 0;
    bool overflow = false;
    uint32_t nTop = 0, nBot = 0; // records the reference's walk dereferences (one per segment looked at)
    // Columns p+1 .. p+remain have the walk of column p with every base advanced by one along its strand: going to a
    // parent, child or paralog keeps the offset inside segments of equal length, so only the reference segment and the
    // segments a parse step lands in can end the run; remain = the fewest bases left (iteration order) in any of them.
    int64_t remain = 0;
    HGX_DEV ColumnWalker(const ColumnParams &p) : P(p) {
    }
    HGX_DEV __forceinline__ const TopRec<C> *top(int g) const {
        return (const TopRec<C> *)P.desc[g].top;
    }
    HGX_DEV __forceinline__ const BotRec<C> *bot(int g) const {
        return (const BotRec<C> *)P.desc[g].bot;
    }
    // (keeping the most recently pushed frame in registers instead of on the scratch stack was measured 17 % slower)
    HGX_DEV __forceinline__ void push(uint32_t kind, int g, int32_t idx, int32_t so, bool rev, int32_t extra) {
        if (sp >= COL_STACK) {
            overflow = true;
            return;
        }
        Frame f;
        f.idx = idx;
        f.so = so;
        f.extra = extra;
        f.meta = kind | ((uint32_t)rev << 3) | ((uint32_t)g << 4);
        stack[sp++] = f;
    }
    HGX_DEV __forceinline__ Frame pop() {
        return stack[--sp];
    }
    // position of a base given its segment and iteration-order offset (halSegmentIterator.cpp:46-52)
    template <typename REC> HGX_DEV __forceinline__ int64_t posOf(const REC *segs, int32_t idx, int32_t so, bool rev) const {
        return !rev ? (int64_t)segs[idx].start + so : (int64_t)segs[idx + 1].start - 1 - so;
    }
    // V: visitor with  void operator()(int genome, int64_t pos, bool rev)
    template <typename V> HGX_DEV __forceinline__ void insert(V &visit, int g, int64_t pos, bool rev) const {
        // colMapInsert (halColumnIterator.cpp:802-812): noAncestors / targets filters
        if constexpr (VisitorWantsRaw<V>::value)
            visit.raw(g, pos);
        if ((!P.noAncestors || P.desc[g].numChildren == 0) && bit(P.targetMask, g))
            visit(g, pos, rev);
    }

    // index of the reference segment (top tiling, or bottom for a genome without one) holding position p
    HGX_DEV __forceinline__ int32_t locate(int64_t p) const {
        const GenomeDesc &RD = P.desc[P.ref];
        int64_t lo = 0, hi;
        if (RD.numTop > 0) {
            const TopRec<C> *T = top(P.ref);
            hi = RD.numTop;
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)T[mid].start <= p)
                    lo = mid;
                else
                    hi = mid;
            }
        } else {
            const BotRec<C> *B = bot(P.ref);
            hi = RD.numBot;
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)B[mid].start <= p)
                    lo = mid;
                else
                    hi = mid;
            }
        }
        return (int32_t)lo;
    }
    template <typename V> HGX_DEV __forceinline__ void run(int64_t p, V &visit) {
        runAt(locate(p), p, visit);
    }
    // the walk of column p, which lies in reference segment seg
    template <typename V> HGX_DEV void runAt(int32_t seg, int64_t p, V &visit) {
        const int R = P.ref;
        const GenomeDesc &RD = P.desc[R];
        sp = 0;
        if (RD.numTop > 0) {
            // recursiveUpdate, top branch (:252-300): toSite, insert, updateParent, updateNextTopDup, updateParseDown
            const TopRec<C> *T = top(R);
            const int32_t t = seg;
            const TopRec<C> tr = T[t];
            if (STATS)
                ++nTop;
            const int32_t so = (int32_t)(p - (int64_t)tr.start);
            remain = (int64_t)T[t + 1].start - 1 - p;
            insert(visit, R, p, false);
            // frames that could only fall through are not pushed (fewer scratch round trips); each frame re-checks
            if (tr.botParse >= 0)
                push(FR_PARSEDOWN, R, t, so, false, 0);
            if (!P.onlyOrthologs && tr.paralogy >= 0)
                push(FR_RING, R, t, so, false, t);
            if (tr.parentEnc >= 0)
                push(FR_UP, R, t, so, false, 0);
        } else {
            // bottom branch (:302-353): the root: insert, then every child
            const BotRec<C> *B = bot(R);
            const int32_t b = seg, so = (int32_t)(p - (int64_t)B[b].start);
            if (STATS)
                ++nBot;
            remain = (int64_t)B[b + 1].start - 1 - p;
            insert(visit, R, p, false);
            for (int i = RD.numChildren - 1; i >= 0; --i)
                if (RD.child[i][b] >= 0)
                    push(FR_CHILD, R, b, so, false, i);
        }
        while (sp > 0) {
            const Frame f = pop();
            const uint32_t kind = f.meta & 7u;
            const bool rev = (f.meta >> 3) & 1u;
            const int g = (int)(f.meta >> 4);
            const GenomeDesc &D = P.desc[g];
            if (kind == FR_UP) {
                // updateParent (:556-605)
                const TopRec<C> tr = top(g)[f.idx];
                if (STATS)
                    ++nTop;
                if (tr.parentEnc >= 0 && D.parent >= 0 && bit(P.scopeMask, D.parent)) {
                    if (STATS)
                        ++nBot;
                    const int pg = D.parent;
                    const GenomeDesc &PD = P.desc[pg];
                    const int32_t b = tr.parentEnc >> 1;
                    // noDupes: only the canonical paralog goes up (mmapTopSegment.cpp:30-40)
                    if (!P.noDupes || (PD.child[D.slotInParent][b] >> 1) == f.idx) {
                        const bool brev = rev ^ ((tr.parentEnc & 1) != 0);
                        insert(visit, pg, posOf(bot(pg), b, f.so, brev), brev);
                        for (int i = PD.numChildren - 1; i >= 0; --i) // siblings, executed after the parse-up branch
                            if (i != D.slotInParent && PD.child[i][b] >= 0)
                                push(FR_CHILD, pg, b, f.so, brev, i);
                        if (PD.parent >= 0)
                            push(FR_PARSEUP, pg, b, f.so, brev, 0);
                    }
                }
            } else if (kind == FR_PARSEUP) {
                // updateParseUp (:683-709): same base seen through the genome's top tiling
                const BotRec<C> *B = bot(g);
                const int32_t tp = B[f.idx].topParse;
                if (tp >= 0) {
                    const int64_t pos = posOf(B, f.idx, f.so, rev);
                    const TopRec<C> *T = top(g);
                    int32_t j = tp;
                    while ((int64_t)T[j + 1].start <= pos)
                        ++j;
                    const TopRec<C> tj = T[j];
                    if (STATS)
                        ++nBot;
                    if (STATS)
                        nTop += (uint32_t)(j - tp + 1);
                    {
                        const int64_t left = !rev ? (int64_t)T[j + 1].start - 1 - pos : pos - (int64_t)tj.start;
                        remain = left < remain ? left : remain;
                    }
                    const int32_t so = !rev ? (int32_t)(pos - (int64_t)tj.start) : (int32_t)((int64_t)T[j + 1].start - 1 - pos);
                    if (!P.onlyOrthologs && tj.paralogy >= 0)
                        push(FR_RING, g, j, so, rev, j);
                    if (tj.parentEnc >= 0)
                        push(FR_UP, g, j, so, rev, 0);
                }
            } else if (kind == FR_CHILD) {
                // updateChild (:607-640)
                const int slot = f.extra;
                const int32_t enc = D.child[slot][f.idx];
                const int cg = D.childGenome[slot];
                if (STATS)
                    ++nBot;
                if (enc >= 0 && bit(P.scopeMask, cg)) {
                    if (STATS)
                        ++nTop;
                    const int32_t t = enc >> 1;
                    const bool crev = rev ^ ((enc & 1) != 0);
                    const TopRec<C> ct = top(cg)[t];
                    insert(visit, cg, !crev ? (int64_t)ct.start + f.so : (int64_t)top(cg)[t + 1].start - 1 - f.so, crev);
                    if (ct.botParse >= 0)
                        push(FR_PARSEDOWN, cg, t, f.so, crev, 0);
                    if (ct.paralogy >= 0)
                        push(FR_RING, cg, t, f.so, crev, t);
                }
            } else if (kind == FR_RING) {
                // updateNextTopDup (:642-681), one ring member per frame: emit the next paralog, walk its subtree,
                // then continue round the ring
                const TopRec<C> *T = top(g);
                const TopRec<C> cur = T[f.idx];
                if (STATS)
                    ++nTop;
                const int32_t first = f.extra;
                const bool startOfRing = f.idx == first;
                bool go = !P.noDupes && cur.paralogy >= 0 && D.parent >= 0 && bit(P.scopeMask, D.parent);
                if (go && !startOfRing)
                    go = cur.paralogy != first; // do/while test (:679-680)
                if (go) {
                    const int32_t nxt = cur.paralogy;
                    const TopRec<C> nr = T[nxt];
                    if (STATS)
                        ++nTop;
                    const bool nrev = rev ^ ((nr.parentEnc & 1) != (cur.parentEnc & 1));
                    insert(visit, g, posOf(T, nxt, f.so, nrev), nrev);
                    if (nr.paralogy >= 0 && nr.paralogy != first)
                        push(FR_RING, g, nxt, f.so, nrev, first);
                    if (nr.botParse >= 0)
                        push(FR_PARSEDOWN, g, nxt, f.so, nrev, 0);
                }
            } else { // FR_PARSEDOWN
                // updateParseDown (:711-744)
                const TopRec<C> *T = top(g);
                const int32_t bp = T[f.idx].botParse;
                if (bp >= 0) {
                    const int64_t pos = posOf(T, f.idx, f.so, rev);
                    const BotRec<C> *B = bot(g);
                    int32_t j = bp;
                    while ((int64_t)B[j + 1].start <= pos)
                        ++j;
                    if (STATS)
                        ++nTop;
                    if (STATS)
                        nBot += (uint32_t)(j - bp + 1);
                    {
                        const int64_t left = !rev ? (int64_t)B[j + 1].start - 1 - pos : pos - (int64_t)B[j].start;
                        remain = left < remain ? left : remain;
                    }
                    const int32_t so = !rev ? (int32_t)(pos - (int64_t)B[j].start) : (int32_t)((int64_t)B[j + 1].start - 1 - pos);
                    for (int i = D.numChildren - 1; i >= 0; --i)
                        if (D.child[i][j] >= 0)
                            push(FR_CHILD, g, j, so, rev, i);
                }
            }
        }
    }
};

// depth visitor: genomes seen (bit set) and bases seen
// (W words of 64 genomes: 4 for alignments of up to 256 genomes, 32 — in scratch — for up to 2048)
template <int W> struct DepthVisitor {
    unsigned long long mask[W];
    uint32_t bases = 0;
    __device__ __forceinline__ DepthVisitor() {
#pragma unroll
        for (int i = 0; i < W; ++i)
            mask[i] = 0;
    }
    __device__ __forceinline__ void operator()(int g, int64_t, bool) {
        mask[g >> 6] |= 1ull << (g & 63);
        ++bases;
    }
    __device__ __forceinline__ int genomes() const {
        int n = 0;
#pragma unroll
        for (int i = 0; i < W; ++i)
            n += (int)__popcll(mask[i]);
        return n;
    }
};

// halAlignmentDepth's per-column value (alignmentDepth/halAlignmentDepth.cpp:258-281): number of genomes with at
// least one base in the column (or, with countDupes, number of bases) minus the reference base.
// Also usable as the row-count pass of the MAF path (countDupes = 2: bases, without the -1).
template <typename C, bool STATS, int W = 4>
__global__ void __launch_bounds__(256) k_column_depth(ColumnParams P, int countMode, int32_t *__restrict__ out) {
    ColumnWalker<C, STATS> w(P);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.count; i += (int64_t)gridDim.x * blockDim.x) {
        DepthVisitor<W> v;
        w.run(P.first + i * P.step, v);
        int32_t val;
        if (countMode == 0)
            val = (int32_t)v.genomes() - 1;
        else if (countMode == 1)
            val = (int32_t)v.bases - 1;
        else
            val = (int32_t)v.bases;
        out[i] = val;
    }
    if (w.overflow)
        *P.error = 1;
    if (STATS && P.derefs) { // wave reduction, one atomic per counter per wave
        uint32_t a = w.nTop, b = w.nBot;
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o);
            b += __shfl_down(b, o);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&P.derefs[0], (unsigned long long)a);
            atomicAdd(&P.derefs[1], (unsigned long long)b);
        }
    }
}

// MAF rows: every reported base of every column, in ColumnMap insertion order
struct ColumnRow {
    int64_t pos;     // genome coordinate
    int32_t genome;
    uint8_t rev;
    char base;       // DnaIterator::getBase: complemented when rev (halDnaIterator.h:131-138)
    uint8_t _pad[2];
};

struct RowVisitor {
    ColumnRow *dst;
    const GenomeDesc *desc;
    uint32_t n = 0;
    HGX_DEV __forceinline__ void operator()(int g, int64_t pos, bool rev) {
        ColumnRow r;
        r.pos = pos;
        r.genome = g;
        r.rev = rev;
        char c = 'N';
        const uint8_t *dna = desc[g].dna;
        if (dna) {
            const uint8_t b = dna[pos >> 1];
            const uint32_t code = (pos & 1) ? (b & 0x0F) : (b >> 4); // dnaUnpack, halCommon.h:187-190
            c = "acgtn\0\0\0ACGTN\0\0"[code];                       // dnaUnpackMap, halCommon.cpp:233-235
            if (rev) { // reverseComplement (halCommon.h:45-75)
                switch (c) {
                case 'A': c = 'T'; break;
                case 'a': c = 't'; break;
                case 'C': c = 'G'; break;
                case 'c': c = 'g'; break;
                case 'G': c = 'C'; break;
                case 'g': c = 'c'; break;
                case 'T': c = 'A'; break;
                case 't': c = 'a'; break;
                default: break;
                }
            }
        }
        r.base = c;
        r._pad[0] = r._pad[1] = 0;
        dst[n++] = r;
    }
};

// cls (optional, --unique: k_column_unique_count): columns of class COL_SKIPPED are not walked (they have no rows)
template <typename C, typename OFF>
__global__ void __launch_bounds__(256) k_column_rows(ColumnParams P, const OFF *__restrict__ rowOffset, ColumnRow *__restrict__ rows,
                                                     const uint8_t *__restrict__ cls = nullptr) {
    ColumnWalker<C> w(P);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.count; i += (int64_t)gridDim.x * blockDim.x) {
        if (cls && cls[i] == 0)
            continue;
        RowVisitor v;
        v.dst = rows + rowOffset[i];
        v.desc = P.desc;
        w.run(P.first + i * P.step, v);
    }
    if (w.overflow)
        *P.error = 1;
}

// ---- hal2maf --unique without the visit cache ----
// With unique set and maxInsertLength == 0 the iterator keeps a visit cache of the REFERENCE genome's bases: every reference
// base a walk reaches right of the range's first column is entered (halColumnIterator.cpp:766-776), nextFreeIndex skips the
// entered ones (:749-764), a walk that meets an entered base is abandoned (:779-800, _break), and hal2maf writes a column only if
// none of its reference bases lies left of the range (isCanonicalOnRef, :210-214; maf/impl/halMafExport.cpp:52-64).  The cache
// is sequential state; what it computes is not.  Every base has at most one parent base and a paralogy ring holds exactly the
// top segments of one parent segment, so the walk of a base reaches the whole tree under the base's topmost in-scope ancestor
// — the same set from whichever of its members it starts (the walk is symmetric under every option: noDupes cuts the same
// edges in both directions, onlyOrthologs never reaches a second base of the reference genome, a scope cuts genomes).  Hence,
// with f the range's first column and R(p) the reference bases of column p's walk:
//   p is walked   <=>  no base of R(p) lies in [f, p)        (otherwise the walk of the smallest such base entered p);
//   p is written  <=>  p is walked and no base of R(p) lies left of f  <=>  p = min R(p);
//   no walk is ever abandoned (a base met a second time would have entered p itself the first time).
// A column that is walked but not written still leaves its sequences as keys in the iterator's column map (resetColMap keeps
// the keys, :822-826), which MafBlock::initBlock turns into empty entries that later columns may join: those columns are
// delivered with their rows, marked.  One lane per column decides this in the pass that counts the rows.
enum : uint8_t { COL_SKIPPED = 0, COL_WRITTEN = 1, COL_KEYS_ONLY = 2 };
struct UniqueVisitor {
    int64_t p, f;   // this column, the range's first column
    int32_t ref;
    uint32_t bases = 0;
    bool inRange = false, leftOfRange = false; // a reference base of the walk in [f, p); one left of f
    __device__ __forceinline__ void raw(int g, int64_t pos) {
        if (g == ref && pos < p) {
            if (pos >= f)
                inRange = true;
            else
                leftOfRange = true;
        }
    }
    __device__ __forceinline__ void operator()(int, int64_t, bool) {
        ++bases;
    }
};
template <> struct VisitorWantsRaw<UniqueVisitor> {
    static constexpr bool value = true;
};
template <typename C>
__global__ void __launch_bounds__(256) k_column_unique_count(ColumnParams P, int64_t rangeFirst, int32_t *__restrict__ cnt, uint8_t *__restrict__ cls) {
    ColumnWalker<C> w(P);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.count; i += (int64_t)gridDim.x * blockDim.x) {
        UniqueVisitor v;
        v.p = P.first + i * P.step;
        v.f = rangeFirst;
        v.ref = P.ref;
        w.run(v.p, v);
        const uint8_t c = v.inRange ? COL_SKIPPED : v.leftOfRange ? COL_KEYS_ONLY : COL_WRITTEN;
        cls[i] = c;
        cnt[i] = c == COL_SKIPPED ? 0 : (int32_t)v.bases;
    }
    if (w.overflow)
        *P.error = 1;
}

// ---- run detection for the MAF writer ----
// A column "continues" its left neighbour when it has the same rows (same genomes and strands, in the same
// insertion order) each advanced by one base along its strand.  Inside such a run the MAF block state machine of
// the reference does nothing but append one character per row, so only the first column of every run ("head")
// needs its rows shipped to the host.
// head[c]: 0 the column continues its left neighbour, 1 a head; with cls (--unique): 2 a column that is not walked, 3 a column
// that is walked but not written (its rows are shipped: they leave keys in the column map) — a written column behind either is a head
static __global__ void __launch_bounds__(256) k_column_heads(const uint32_t *__restrict__ rowOffset, const ColumnRow *__restrict__ rows, int64_t count,
                                                      uint8_t *__restrict__ head, uint32_t *__restrict__ headRows /* rows of heads, else 0 */,
                                                      const uint8_t *__restrict__ cls = nullptr) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < count; c += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t a = rowOffset[c], n = rowOffset[c + 1] - a;
        if (cls && cls[c] != COL_WRITTEN) {
            head[c] = cls[c] == COL_SKIPPED ? 2 : 3;
            headRows[c] = cls[c] == COL_SKIPPED ? 0 : n;
            continue;
        }
        bool isHead = c == 0 || (cls && cls[c - 1] != COL_WRITTEN);
        if (!isHead) {
            const uint32_t pa = rowOffset[c - 1];
            isHead = (a - pa) != n;
            for (uint32_t k = 0; k < n && !isHead; ++k) {
                const ColumnRow r = rows[a + k], q = rows[pa + k];
                isHead = r.genome != q.genome || r.rev != q.rev || r.pos != (q.rev ? q.pos - 1 : q.pos + 1);
            }
        }
        head[c] = isHead ? 1 : 0;
        headRows[c] = isHead ? n : 0;
    }
}

static __global__ void __launch_bounds__(256) k_gather_head_rows(const uint32_t *__restrict__ rowOffset, const ColumnRow *__restrict__ rows, int64_t count,
                                                          const uint8_t *__restrict__ head, const uint32_t *__restrict__ headOffset,
                                                          ColumnRow *__restrict__ out) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < count; c += (int64_t)gridDim.x * blockDim.x) {
        if (!(head[c] & 1))
            continue;
        const uint32_t a = rowOffset[c], n = rowOffset[c + 1] - a, o = headOffset[c];
        for (uint32_t k = 0; k < n; ++k)
            out[o + k] = rows[a + k];
    }
}

// ---------------------------------------------------------------------------------------------
// Run-compressed column engine.  One lane owns one reference SEGMENT and walks only the first column of every run
// (ColumnWalker::remain tells where the next run starts); the columns inside a run differ from its first column by
// a shift, so depth is constant along a run and MAF rows advance by one base per column.  This is "the closure
// computed once per reference piece" of SURVEY 8(d): ~24x fewer walks than one per column on the cfg2 alignment.
struct LongRun { // a run too long for its lane to fill alone
    int64_t first, count; // output index range
    int32_t value, _pad;
};
static constexpr int64_t LANE_FILL_MAX = 256;

template <int W> __device__ __forceinline__ int32_t depth_value(const DepthVisitor<W> &v, int countMode) {
    if (countMode == 0)
        return (int32_t)v.genomes() - 1;
    if (countMode == 1)
        return (int32_t)v.bases - 1;
    return (int32_t)v.bases;
}

template <typename W> __device__ __forceinline__ void ref_segment_bounds(const W &w, int32_t s, int64_t &lo, int64_t &hi) {
    const GenomeDesc &RD = w.P.desc[w.P.ref];
    if (RD.numTop > 0) {
        lo = (int64_t)w.top(w.P.ref)[s].start;
        hi = (int64_t)w.top(w.P.ref)[s + 1].start - 1;
    } else {
        lo = (int64_t)w.bot(w.P.ref)[s].start;
        hi = (int64_t)w.bot(w.P.ref)[s + 1].start - 1;
    }
}

// out[i] for the columns first + i*step (i < count); segFirst .. segFirst+segCount-1 are the reference segments they lie in
template <typename C, bool STATS, int W = 4>
__global__ void __launch_bounds__(256) k_depth_runs(ColumnParams P, int countMode, int32_t segFirst, int64_t segCount, int32_t *__restrict__ out,
                                                    LongRun *__restrict__ longRuns, unsigned long long *__restrict__ longCount,
                                                    unsigned long long longCap) {
    ColumnWalker<C, STATS> w(P);
    const int64_t lastPos = P.first + (P.count - 1) * P.step;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < segCount; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = segFirst + (int32_t)k;
        int64_t lo, hi;
        ref_segment_bounds(w, s, lo, hi);
        int64_t p = lo > P.first ? lo : P.first;
        const int64_t end = hi < lastPos ? hi : lastPos;
        if (P.step > 1) { // first sampled column at or after p
            const int64_t r = (p - P.first) % P.step;
            if (r)
                p += P.step - r;
        }
        while (p <= end) {
            DepthVisitor<W> v;
            w.runAt(s, p, v);
            const int32_t val = depth_value(v, countMode);
            const int64_t q = p + w.remain < end ? p + w.remain : end; // last column of the run (inside this segment)
            const int64_t i0 = (p - P.first) / P.step, i1 = (q - P.first) / P.step;
            const int64_t n = i1 - i0 + 1;
            if (n <= LANE_FILL_MAX) {
                for (int64_t i = i0; i <= i1; ++i)
                    out[i] = val;
            } else {
                const unsigned long long slot = atomicAdd(longCount, 1ull);
                if (slot < longCap) {
                    LongRun lr;
                    lr.first = i0;
                    lr.count = n;
                    lr.value = val;
                    lr._pad = 0;
                    longRuns[slot] = lr;
                }
            }
            p = P.first + (i1 + 1) * P.step;
        }
    }
    if (w.overflow)
        *P.error = 1;
    if (STATS && P.derefs) {
        uint32_t a = w.nTop, b = w.nBot;
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o);
            b += __shfl_down(b, o);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&P.derefs[0], (unsigned long long)a);
            atomicAdd(&P.derefs[1], (unsigned long long)b);
        }
    }
}

static __global__ void __launch_bounds__(256) k_fill_long_runs(const LongRun *__restrict__ runs, const unsigned long long *__restrict__ nRuns,
                                                               int32_t *__restrict__ out) {
    const unsigned long long n = *nRuns;
    for (unsigned long long r = blockIdx.x; r < n; r += gridDim.x) {
        const LongRun lr = runs[r];
        for (int64_t i = threadIdx.x; i < lr.count; i += blockDim.x)
            out[lr.first + i] = lr.value;
    }
}

// ---- depth by two sweeps over the tree (hgx_columns.hip: columnsDepthSweep) ----
// Every base has at most one parent base, so the column of a base — everything the walk of recursiveUpdate reaches
// (api/impl/halColumnIterator.cpp:246-355: up to the parent, across to the siblings and paralogs, down into every child) — is the
// whole descendant tree of the base's topmost ancestor.  With S_x[p] = the genomes below base p of genome x (a bit set; x's own
// bit if x counts), the depth of a column is the size of S at that ancestor.  Two streaming sweeps, one track entry per base:
//   bottom-up  S_x over a bottom segment = own bit | the S of every top segment of every child that hangs under it (the
//              child slot's segment and its paralogy ring), read in the child's orientation;
//   top-down   A_x[p] = size of S at p's topmost ancestor: copied from the parent's A where the top segment has a parent,
//              the size of S_x[p] where it has none (or x is the top of the walk's scope).
// halAlignmentDepth's value is A - 1 on the reference (alignmentDepth/halAlignmentDepth.cpp:258-281).  SUM = true counts
// bases instead of genomes (--countDupes): the same sweeps with sums.  One wavefront per segment, lanes along its bases.
template <typename M, bool SUM> __device__ __forceinline__ M track_join(M a, M b) {
    return SUM ? (M)(a + b) : (M)(a | b);
}
// A genome's set holds only the genomes of its own subtree, numbered from the subtree's first counted genome (post-order: a
// subtree's genomes are consecutive), in a word just wide enough for them — most genomes of a large tree have a handful of
// descendants, and the sweeps are bound by the tracks' bytes (the 50-genome alignment: 8 bytes a base for every genome moved
// 69 GB, most of it for sets with three members).  A parent reads a child's word, of the child's width, and moves it up by the
// child's place in its own numbering (shift).
struct SweepChild {
    const int32_t *enc; // the parent's child link array for this slot
    const void *top;    // the child's TopRec table
    const void *track;  // the child's S, or null: the same value on every base (a genome without in-scope children)
    long long constant; // that value, in the parent's numbering
    int wlog;           // log2 of the bytes of a word of the child's track
    int shift;          // the child's first genome in the parent's numbering (0 for sums)
};
static constexpr int SWEEP_MAX_CHILDREN = 8;
struct SweepChildren {
    SweepChild c[SWEEP_MAX_CHILDREN];
    int n;
    int noRing; // --noDupes: a child slot's segment without its paralogy ring (updateNextTopDup is not run: halColumnIterator.cpp:642-681)
};
// sixteen lanes per bottom segment — four segments' dependent loads (segment bounds, child link, child record, ring links) are
// in flight per wavefront instead of one —, and every lane moves 8 bytes of consecutive bases of the track (16 of 64-bit sets) as one
// word (a child in the other orientation: the word at the mirrored place, its elements taken back to front).  The tracks start
// where their segments start: the words are not aligned (global accesses need not be); the last word of a segment goes
// element by element.
#ifndef HGX_SWEEP_LPS_LOG
#define HGX_SWEEP_LPS_LOG(V) ((V) >= 8 ? 3 : 4)
#endif
template <typename M> struct SweepVec {
    // (sixteen bytes a lane for 32- and 64-bit words — round 6: the sweeps are bound by the instructions a lane spends beside its
    // loads, and twice the bytes a lane are half of those a byte: --countDupes at config 2 4.78 -> 3.95 ms; with 16-bit sets the
    // eight elements cost registers instead, 56 -> 76 VGPRs, and the root's launch got slower)
    static constexpr int N = sizeof(M) >= 8 ? 2 : sizeof(M) >= 4 ? 4 : 8 / (int)sizeof(M);
    M e[N];
};
template <typename T, int N> struct SweepElems {
    T e[N];
};
template <typename T> __device__ __forceinline__ T sweep_load(const void *p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}
template <typename T> __device__ __forceinline__ void sweep_store(void *p, const T &v) {
    __builtin_memcpy(p, &v, sizeof(T));
}
// the bases o .. o + V - 1 of a parent segment of `len` bases under child segment `tr`, from the child's track T (words of MC)
template <typename C, typename M, typename MC, bool SUM>
__device__ __forceinline__ void sweep_join_child(SweepVec<M> &v, const MC *__restrict__ T, const TopRec<C> &tr, C len, C o, bool whole,
                                                 int shift) {
    constexpr int V = SweepVec<M>::N;
    const MC *base = T + (int64_t)tr.start;
    const bool rev = (tr.parentEnc & 1) != 0;
    if (whole) {
        const SweepElems<MC, V> x = sweep_load<SweepElems<MC, V>>(base + (rev ? len - o - V : o));
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const MC c = x.e[rev ? V - 1 - j : j];
            v.e[j] = track_join<M, SUM>(v.e[j], SUM ? (M)c : (M)((unsigned long long)c << shift));
        }
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j)
            if (o + j < len) {
                const MC c = base[rev ? len - 1 - o - j : o + j];
                v.e[j] = track_join<M, SUM>(v.e[j], SUM ? (M)c : (M)((unsigned long long)c << shift));
            }
    }
}
// OUT: what is stored — the sets themselves (M), or, for the genome at the top of a depth request's scope, their sizes as bytes
// (round 6: nobody reads that genome's sets but to count them — k_sweep_top, the first step of k_sweep_down —, and with 64-bit sets
// they are eight bytes a base written and read again where one is enough; not with `accumulate`: a later launch adds to the sets)
template <typename C, typename M, bool SUM, typename OUT = M>
static __global__ void __launch_bounds__(256) k_sweep_up(const BotRec<C> *__restrict__ bot, int64_t numBot, SweepChildren ch, M own, int accumulate,
                                                         OUT *__restrict__ S) {
    constexpr int V = SweepVec<M>::N;
    constexpr bool SIZES = !std::is_same<OUT, M>::value;
    // (lanes per segment: a round of them covers 64 bases or more — eight lanes with byte-wide sets, so that eight segments'
    // chains of dependent loads are in flight per wavefront)
    constexpr int LPS_LOG = HGX_SWEEP_LPS_LOG(V);
    constexpr int LPS = 1 << LPS_LOG;
    const int sub = (int)(threadIdx.x & (LPS - 1));
    const int64_t groupsTotal = ((int64_t)gridDim.x * blockDim.x) >> LPS_LOG;
    for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> LPS_LOG; b < numBot; b += groupsTotal) {
        // (offsets inside a segment in the tables' own width — with 32-bit tables a lane's address arithmetic is 32-bit: the sweeps
        // are bound by the instructions a lane spends on its few bytes, round 6)
        const int64_t start = (int64_t)bot[b].start;
        const C len = (C)(bot[b + 1].start - bot[b].start);
        for (C o0 = 0; o0 < len; o0 += LPS * V) {
            C o = o0 + (C)(sub * V); // this lane's bases: o .. o + V - 1
            if (o >= len)
                continue;
            // (the lane at the segment's end takes the segment's last V bases — some of them its neighbour's as well, which come out
            // the same: element by element it was the one lane the other fifteen waited for.  Not when sums are added to what an
            // earlier launch left (more than SWEEP_MAX_CHILDREN children, --countDupes): a base covered twice — by this lane and by
            // the last lane of the round before, which has stored already — would get this launch's children added twice)
            if (!(SUM && accumulate) && o + V > len && len >= V)
                o = len - V;
            const bool whole = o + V <= len;
            SweepVec<M> v;
            if constexpr (SIZES) {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    v.e[j] = own;
            } else if (accumulate && whole) { // (more than SWEEP_MAX_CHILDREN children: several launches)
                v = sweep_load<SweepVec<M>>(S + start + o);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    v.e[j] = accumulate && o + j < len ? S[start + o + j] : own;
            }
            for (int k = 0; k < ch.n; ++k) {
                const int32_t enc = ch.c[k].enc[b];
                if (enc < 0)
                    continue;
                const TopRec<C> *top = (const TopRec<C> *)ch.c[k].top;
                const void *T = ch.c[k].track;
                if (!T && !SUM) { // a child without tracks of its own: every base below carries the same set
                    const M cst = (M)ch.c[k].constant;
#pragma unroll
                    for (int j = 0; j < V; ++j)
                        v.e[j] = track_join<M, SUM>(v.e[j], cst);
                    continue;
                }
                const int32_t t0 = enc >> 1;
                int32_t t = t0;
                do { // the slot's segment and its paralogy ring (updateChild + updateNextTopDup, :607-681)
                    const TopRec<C> tr = top[t];
                    if (T) {
                        const int shift = ch.c[k].shift;
                        switch (ch.c[k].wlog) {
                        case 0:
                            sweep_join_child<C, M, uint8_t, SUM>(v, (const uint8_t *)T, tr, len, o, whole, shift);
                            break;
                        case 1:
                            sweep_join_child<C, M, uint16_t, SUM>(v, (const uint16_t *)T, tr, len, o, whole, shift);
                            break;
                        case 2:
                            sweep_join_child<C, M, uint32_t, SUM>(v, (const uint32_t *)T, tr, len, o, whole, shift);
                            break;
                        default:
                            sweep_join_child<C, M, unsigned long long, SUM>(v, (const unsigned long long *)T, tr, len, o, whole, shift);
                            break;
                        }
                    } else { // (sums: one per ring member)
#pragma unroll
                        for (int j = 0; j < V; ++j)
                            v.e[j] = track_join<M, SUM>(v.e[j], (M)ch.c[k].constant);
                    }
                    t = ch.noRing ? -1 : tr.paralogy;
                } while (t >= 0 && t != t0);
            }
            if constexpr (SIZES) {
                SweepElems<OUT, V> w;
#pragma unroll
                for (int j = 0; j < V; ++j)
                    w.e[j] = (OUT)(SUM ? (long long)v.e[j] : (long long)__popcll((unsigned long long)v.e[j]));
                if (whole) {
                    sweep_store(S + start + o, w);
                } else {
#pragma unroll
                    for (int j = 0; j < V; ++j)
                        if (o + j < len)
                            S[start + o + j] = w.e[j];
                }
            } else if (whole) {
                sweep_store(S + start + o, v);
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (o + j < len)
                        S[start + o + j] = v.e[j];
            }
        }
    }
}
// k_sweep_up for byte-wide sets (a genome with at most eight counted genomes below it: most genomes of a large tree), round 6.
// What the launch waits for is not bytes: a wavefront's round — the segments' bounds and links, the children's records, their
// tracks, the store — is a few memory round trips whatever it moves (loads and stores are counted by one in-order counter on
// this hardware: 3.6 us a round, measured on the launch that only stores), and a round moved eight segments of 64 bases.  So a
// round moves more: U segments a lane group, their loads asked for together level by level, and sixteen bases a lane — a
// segment of up to 128 bases in one trip.  (Asking for the NEXT round's links and records beside this round's tracks, a
// pipeline over rounds, changed nothing: profiles/r06x_*.)  A lane's sixteen bases are two 64-bit words from the load to the
// store: a child in the other orientation by byte swaps, its place in the parent's numbering by a masked shift.
struct SweepW2 {
    unsigned long long a, b;
};
template <typename O> __device__ __forceinline__ SweepW2 sweep_bytes16(const uint8_t *__restrict__ base, bool rev, O len, O o, bool whole) {
    SweepW2 x;
    if (whole) {
        x = sweep_load<SweepW2>(base + (rev ? len - o - 16 : o));
        if (rev) {
            const unsigned long long t = __builtin_bswap64(x.a);
            x.a = __builtin_bswap64(x.b);
            x.b = t;
        }
    } else {
        x.a = x.b = 0;
#pragma nounroll
        for (int j = 0; j < 16 && o + j < len; ++j) {
            const unsigned long long v = (unsigned long long)base[rev ? len - 1 - o - j : o + j] << (8 * (j & 7));
            if (j < 8)
                x.a |= v;
            else
                x.b |= v;
        }
    }
    return x;
}
// a segment of a lane group's round (its fields as scalars and U of them by name, not as arrays: indexed arrays of them the
// compiler kept in LDS)
template <typename C> struct SweepSeg {
    C start, len, o; // (offsets in the tables' own width: with 32-bit tables the address arithmetic of a lane is 32-bit)
    bool act, whole;
    SweepW2 pv, x;
    int32_t enc;
    TopRec<C> tr;
};
#define HGX_SEGS(X) X(s0) X(s1)
// per-byte population counts of eight bytes at once
__device__ __forceinline__ unsigned long long sweep_popcount_bytes(unsigned long long x) {
    x = x - ((x >> 1) & 0x5555555555555555ull);
    x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
    return (x + (x >> 4)) & 0x0F0F0F0F0F0F0F0Full;
}
// SIZES: the sets' sizes are stored, not the sets (the genome at the top of a depth request's scope: k_sweep_up's OUT)
template <typename C, int U, bool SIZES = false>
static __global__ void __launch_bounds__(256) k_sweep_up_bytes(const BotRec<C> *__restrict__ bot, int64_t numBot, SweepChildren ch, uint8_t own,
                                                               int accumulate, uint8_t *__restrict__ S) {
    static_assert(U == 2, "HGX_SEGS names the segments");
    const int sub = (int)(threadIdx.x & 7);
    const int64_t G = ((int64_t)gridDim.x * blockDim.x) >> 3;
    const unsigned long long ones = 0x0101010101010101ull;
    const int nKids = ch.n, noRing = ch.noRing;
    for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; b < numBot; b += (int64_t)U * G) {
        SweepSeg<C> s0, s1;
        const int64_t b_s0 = b, b_s1 = b + G;
#define X(s)                                                                                                                                 \
    s.start = s.len = 0;                                                                                                                     \
    if (b_##s < numBot) {                                                                                                                    \
        s.start = bot[b_##s].start;                                                                                                          \
        s.len = (C)(bot[b_##s + 1].start - s.start);                                                                                         \
    }
        HGX_SEGS(X)
#undef X
        const C maxLen = s0.len > s1.len ? s0.len : s1.len;
        for (C o0 = 0; o0 < maxLen; o0 += 128) {
#define X(s)                                                                                                                                 \
    s.o = o0 + (C)(sub * 16);                                                                                                                \
    s.act = s.o < s.len;                                                                                                                     \
    if (s.act && s.o + 16 > s.len && s.len >= 16)                                                                                            \
        s.o = s.len - 16; /* (the lane at the segment's end takes its last sixteen bases: unions may be made twice) */                      \
    s.whole = s.o + 16 <= s.len;                                                                                                             \
    s.pv.a = s.pv.b = ones * (unsigned long long)own;                                                                                        \
    if (accumulate && s.act)                                                                                                                 \
        s.pv = sweep_bytes16<C>(S + (int64_t)s.start, false, s.len, s.o, s.whole);
            HGX_SEGS(X)
#undef X
            for (int k = 0; k < nKids; ++k) {
                const int32_t *encs = ch.c[k].enc;
                const uint8_t *T = (const uint8_t *)ch.c[k].track;
                const TopRec<C> *top = (const TopRec<C> *)ch.c[k].top;
                const int shift = ch.c[k].shift;
                const unsigned long long keep = ones * (unsigned long long)(0xFFu >> shift), cst = ones * (unsigned long long)(uint8_t)ch.c[k].constant;
#define X(s) s.enc = s.act ? encs[b_##s] : -1;
                HGX_SEGS(X)
#undef X
                if (!T) { // (a child without a track: the same set on every base below it)
#define X(s)                                                                                                                                 \
    if (s.enc >= 0) {                                                                                                                        \
        s.pv.a |= cst;                                                                                                                       \
        s.pv.b |= cst;                                                                                                                       \
    }
                    HGX_SEGS(X)
#undef X
                    continue;
                }
#define X(s)                                                                                                                                 \
    s.tr.start = 0;                                                                                                                          \
    s.tr.parentEnc = 0;                                                                                                                      \
    s.tr.paralogy = -1;                                                                                                                      \
    if (s.enc >= 0)                                                                                                                          \
        s.tr = top[s.enc >> 1];
                HGX_SEGS(X)
#undef X
#define X(s)                                                                                                                                 \
    s.x.a = s.x.b = 0;                                                                                                                       \
    if (s.enc >= 0)                                                                                                                          \
        s.x = sweep_bytes16<C>(T + (int64_t)s.tr.start, (s.tr.parentEnc & 1) != 0, s.len, s.o, s.whole);
                HGX_SEGS(X)
#undef X
                // (behind the slot's segment the rest of its paralogy ring — updateNextTopDup, halColumnIterator.cpp:642-681: rare, link by link)
#define X(s)                                                                                                                                 \
    s.pv.a |= (s.x.a & keep) << shift;                                                                                                       \
    s.pv.b |= (s.x.b & keep) << shift;                                                                                                       \
    if (s.enc >= 0 && !noRing && s.tr.paralogy >= 0) {                                                                                       \
        const int32_t t0 = s.enc >> 1;                                                                                                       \
        for (int32_t t = s.tr.paralogy; t >= 0 && t != t0;) {                                                                                \
            const TopRec<C> r = top[t];                                                                                                      \
            const SweepW2 y = sweep_bytes16<C>(T + (int64_t)r.start, (r.parentEnc & 1) != 0, s.len, s.o, s.whole);                           \
            s.pv.a |= (y.a & keep) << shift;                                                                                                 \
            s.pv.b |= (y.b & keep) << shift;                                                                                                 \
            t = r.paralogy;                                                                                                                  \
        }                                                                                                                                    \
    }
                HGX_SEGS(X)
#undef X
            }
#define X(s)                                                                                                                                 \
    if (s.act) {                                                                                                                             \
        if (SIZES) {                                                                                                                         \
            s.pv.a = sweep_popcount_bytes(s.pv.a);                                                                                           \
            s.pv.b = sweep_popcount_bytes(s.pv.b);                                                                                           \
        }                                                                                                                                    \
        if (s.whole) {                                                                                                                       \
            sweep_store(S + (int64_t)s.start + s.o, s.pv);                                                                                   \
        } else {                                                                                                                             \
            _Pragma("nounroll") for (int j = 0; j < 16 && s.o + j < s.len; ++j)                                                              \
                S[(int64_t)s.start + s.o + j] = (uint8_t)((j < 8 ? s.pv.a : s.pv.b) >> (8 * (j & 7)));                                      \
        }                                                                                                                                    \
    }
            HGX_SEGS(X)
#undef X
        }
    }
}
#undef HGX_SEGS
// ---- k_sweep_up_bytes' form for sets of 2, 4 and 8 bytes (round 6) ----
// A lane's sixteen bytes are N = 16 / W bases of W-byte sets in two 64-bit words; a child's track has words of CW <= W bytes: its
// N elements are loaded as N * CW bytes, turned round when the child lies the other way, widened to W bytes and moved to the child's
// place in the parent's numbering by one masked shift.  8 * W lanes a segment (128 bases a round whatever the width), two segments
// a lane group.
template <int CW> __device__ __forceinline__ unsigned long long sweep_reverse_elems64(unsigned long long x) { // the 8 / CW elements of a word, back to front
    if (CW == 8)
        return x;
    if (CW == 4)
        return (x >> 32) | (x << 32);
    x = __builtin_bswap64(x);
    if (CW == 2)
        x = ((x & 0xFF00FF00FF00FF00ull) >> 8) | ((x & 0x00FF00FF00FF00FFull) << 8);
    return x;
}
template <int W> __device__ __forceinline__ unsigned long long sweep_replicate(unsigned long long v) { // a W-byte value in every W bytes of a word
    return W == 1 ? 0x0101010101010101ull * (v & 0xFFull) : W == 2 ? 0x0001000100010001ull * (v & 0xFFFFull) : W == 4 ? 0x0000000100000001ull * (v & 0xFFFFFFFFull) : v;
}
// the N elements of a child's track (words of CW bytes; ts: the child segment's start) under the parent's bases o .. o + N - 1, in
// the parent's order, widened to W bytes
template <int W, int CW, typename O>
__device__ __forceinline__ SweepW2 sweep_fetch_words(const uint8_t *__restrict__ T, int64_t ts, bool rev, O len, O o, bool whole) {
    constexpr int N = 16 / W, CB = N * CW; // bases a lane; the child's bytes under them
    SweepW2 x;
    x.a = x.b = 0;
    if (!whole) {
#pragma nounroll
        for (int j = 0; j < N && o + j < len; ++j) {
            const int64_t at = (ts + (rev ? (int64_t)len - 1 - o - j : (int64_t)o + j)) * CW;
            unsigned long long e = 0;
            __builtin_memcpy(&e, T + at, CW);
            if (j < N / 2)
                x.a |= W == 8 ? e : e << ((8 * W * j) & 63);
            else
                x.b |= W == 8 ? e : e << ((8 * W * (j - N / 2)) & 63);
        }
        return x;
    }
    const uint8_t *at = T + (ts + (rev ? (int64_t)len - o - N : (int64_t)o)) * CW;
    if (CB == 16) { // (CW == W: the child's words are the parent's)
        x = sweep_load<SweepW2>(at);
        if (rev) {
            const unsigned long long t = sweep_reverse_elems64<CW>(x.a);
            x.a = sweep_reverse_elems64<CW>(x.b);
            x.b = t;
        }
        return x;
    }
    unsigned long long c = 0; // the child's N elements, CB <= 8 bytes
    __builtin_memcpy(&c, at, CB);
    if (rev) {
        c = sweep_reverse_elems64<CW>(c);
        c >>= 8 * (8 - CB); // (the elements were at the word's low end: turned round they are at its high end)
    }
    const unsigned long long em = CW == 8 ? ~0ull : ((1ull << (8 * CW)) - 1ull);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const unsigned long long e = (c >> (8 * CW * j)) & em;
        if (j < N / 2)
            x.a |= e << ((8 * W * j) & 63);
        else
            x.b |= e << ((8 * W * (j - N / 2)) & 63);
    }
    return x;
}
template <int W> __device__ __forceinline__ unsigned long long sweep_shift_words(unsigned long long x, int shift) {
    if (W == 8)
        return x << shift;
    const unsigned long long all = W == 1 ? 0xFFull : W == 2 ? 0xFFFFull : 0xFFFFFFFFull;
    return (x & sweep_replicate<W>(all >> shift)) << shift;
}
// the sizes of a lane's N sets as N bytes (the low N bytes of the result)
template <int W> __device__ __forceinline__ unsigned long long sweep_sizes_words(const SweepW2 &v) {
    if (W == 8)
        return (unsigned long long)__popcll(v.a) | ((unsigned long long)__popcll(v.b) << 8);
    if (W == 4)
        return (unsigned long long)__popc((unsigned)v.a) | ((unsigned long long)__popc((unsigned)(v.a >> 32)) << 8) |
               ((unsigned long long)__popc((unsigned)v.b) << 16) | ((unsigned long long)__popc((unsigned)(v.b >> 32)) << 24);
    // W == 2: per-byte counts, pairs of them added, the four 16-bit sums of a word packed into four bytes
    unsigned long long r = 0;
    const unsigned long long w[2] = {v.a, v.b};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned long long t = sweep_popcount_bytes(w[h]);
        t = (t + (t >> 8)) & 0x00FF00FF00FF00FFull;
        t = (t | (t >> 8)) & 0x0000FFFF0000FFFFull;
        t = (t | (t >> 16)) & 0x00000000FFFFFFFFull;
        r |= t << (32 * h);
    }
    return r;
}
// two 32-bit sums side by side in a word, added without a carry between them
__device__ __forceinline__ unsigned long long sweep_add32x2(unsigned long long a, unsigned long long b) {
    return (unsigned long long)((uint32_t)a + (uint32_t)b) | ((unsigned long long)((uint32_t)(a >> 32) + (uint32_t)(b >> 32)) << 32);
}
// SUM (W = 4): the words are counts of bases, joined by adding (--countDupes, hal2maf's tracks): a child's words are counts too, a child
// without a track adds its constant once for the slot's segment and once for every other member of its paralogy ring
template <typename C, int W, bool SIZES, bool SUM = false>
static __global__ void __launch_bounds__(256) k_sweep_up_words(const BotRec<C> *__restrict__ bot, int64_t numBot, SweepChildren ch, unsigned long long own,
                                                               int accumulate, uint8_t *__restrict__ S) {
    static_assert(W == 2 || W == 4 || W == 8, "byte-wide sets: k_sweep_up_bytes");
    static_assert(!SUM || (W == 4 && !SIZES), "sums are 32-bit counts");
    constexpr int N = 16 / W;                      // bases a lane
    constexpr int LPS_LOG = W == 2 ? 4 : W == 4 ? 5 : 6; // lanes a segment: 128 bases a round
    constexpr int LPS = 1 << LPS_LOG;
    const int sub = (int)(threadIdx.x & (LPS - 1));
    const int64_t G = ((int64_t)gridDim.x * blockDim.x) >> LPS_LOG;
    const int nKids = ch.n, noRing = ch.noRing;
    const unsigned long long ownAll = sweep_replicate<W>(own);
#define HGX_SEGS(X) X(s0) X(s1)
    for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> LPS_LOG; b < numBot; b += 2 * G) {
        SweepSeg<C> s0, s1;
        const int64_t b_s0 = b, b_s1 = b + G;
#define X(s)                                                                                                                                 \
    s.start = s.len = 0;                                                                                                                     \
    if (b_##s < numBot) {                                                                                                                    \
        s.start = bot[b_##s].start;                                                                                                          \
        s.len = (C)(bot[b_##s + 1].start - s.start);                                                                                         \
    }
        HGX_SEGS(X)
#undef X
        const C maxLen = s0.len > s1.len ? s0.len : s1.len;
        for (C o0 = 0; o0 < maxLen; o0 += LPS * N) {
#define X(s)                                                                                                                                 \
    s.o = o0 + (C)(sub * N);                                                                                                                 \
    s.act = s.o < s.len;                                                                                                                     \
    if (s.act && !(SUM && accumulate) && s.o + N > s.len && s.len >= N)                                                                      \
        s.o = s.len - N; /* (the lane at the segment's end takes its last N bases: unions may be made twice — sums added to what an      \
                            earlier launch left may not: k_sweep_up) */                                                                     \
    s.whole = s.o + N <= s.len;                                                                                                              \
    s.pv.a = s.pv.b = ownAll;                                                                                                                \
    if (accumulate && s.act)                                                                                                                 \
        s.pv = sweep_fetch_words<W, W, C>(S, (int64_t)s.start, false, s.len, s.o, s.whole);
            HGX_SEGS(X)
#undef X
            for (int k = 0; k < nKids; ++k) {
                const int32_t *encs = ch.c[k].enc;
                const uint8_t *T = (const uint8_t *)ch.c[k].track;
                const TopRec<C> *top = (const TopRec<C> *)ch.c[k].top;
                const int shift = ch.c[k].shift, wlog = ch.c[k].wlog;
                const unsigned long long cst = sweep_replicate<W>((unsigned long long)ch.c[k].constant);
#define X(s) s.enc = s.act ? encs[b_##s] : -1;
                HGX_SEGS(X)
#undef X
                if (!T && !SUM) { // (a child without a track: the same set on every base below it)
#define X(s)                                                                                                                                 \
    if (s.enc >= 0) {                                                                                                                        \
        s.pv.a |= cst;                                                                                                                       \
        s.pv.b |= cst;                                                                                                                       \
    }
                    HGX_SEGS(X)
#undef X
                    continue;
                }
#define X(s)                                                                                                                                 \
    s.tr.start = 0;                                                                                                                          \
    s.tr.parentEnc = 0;                                                                                                                      \
    s.tr.paralogy = -1;                                                                                                                      \
    if (s.enc >= 0)                                                                                                                          \
        s.tr = top[s.enc >> 1];
                HGX_SEGS(X)
#undef X
                // a child's words are as wide as the parent's or narrower (its subtree is part of the parent's)
#define HGX_FETCH(s, TR)                                                                                                                     \
    (wlog == 0 ? sweep_fetch_words<W, 1, C>(T, (int64_t)(TR).start, ((TR).parentEnc & 1) != 0, s.len, s.o, s.whole)                          \
     : wlog == 1 || W == 2 ? sweep_fetch_words<W, 2, C>(T, (int64_t)(TR).start, ((TR).parentEnc & 1) != 0, s.len, s.o, s.whole)             \
     : wlog == 2 || W == 4 ? sweep_fetch_words<W, (W >= 4 ? 4 : W), C>(T, (int64_t)(TR).start, ((TR).parentEnc & 1) != 0, s.len, s.o, s.whole) \
                           : sweep_fetch_words<W, W, C>(T, (int64_t)(TR).start, ((TR).parentEnc & 1) != 0, s.len, s.o, s.whole))
#define X(s)                                                                                                                                 \
    s.x.a = s.x.b = 0;                                                                                                                       \
    if (s.enc >= 0) {                                                                                                                        \
        if (SUM && !T)                                                                                                                       \
            s.x.a = s.x.b = cst;                                                                                                             \
        else                                                                                                                                 \
            s.x = HGX_FETCH(s, s.tr);                                                                                                        \
    }
                HGX_SEGS(X)
#undef X
                // (behind the slot's segment the rest of its paralogy ring — updateNextTopDup, halColumnIterator.cpp:642-681: rare, link by link)
#define X(s)                                                                                                                                 \
    if (SUM) {                                                                                                                               \
        s.pv.a = sweep_add32x2(s.pv.a, s.x.a);                                                                                               \
        s.pv.b = sweep_add32x2(s.pv.b, s.x.b);                                                                                               \
    } else {                                                                                                                                 \
        s.pv.a |= sweep_shift_words<W>(s.x.a, shift);                                                                                        \
        s.pv.b |= sweep_shift_words<W>(s.x.b, shift);                                                                                        \
    }                                                                                                                                        \
    if (s.enc >= 0 && !noRing && s.tr.paralogy >= 0) {                                                                                       \
        const int32_t t0 = s.enc >> 1;                                                                                                       \
        for (int32_t t = s.tr.paralogy; t >= 0 && t != t0;) {                                                                                \
            const TopRec<C> r = top[t];                                                                                                      \
            SweepW2 y;                                                                                                                       \
            if (SUM && !T)                                                                                                                   \
                y.a = y.b = cst;                                                                                                             \
            else                                                                                                                             \
                y = HGX_FETCH(s, r);                                                                                                         \
            if (SUM) {                                                                                                                       \
                s.pv.a = sweep_add32x2(s.pv.a, y.a);                                                                                         \
                s.pv.b = sweep_add32x2(s.pv.b, y.b);                                                                                         \
            } else {                                                                                                                         \
                s.pv.a |= sweep_shift_words<W>(y.a, shift);                                                                                  \
                s.pv.b |= sweep_shift_words<W>(y.b, shift);                                                                                  \
            }                                                                                                                                \
            t = r.paralogy;                                                                                                                  \
        }                                                                                                                                    \
    }
                HGX_SEGS(X)
#undef X
#undef HGX_FETCH
            }
#define X(s)                                                                                                                                 \
    if (s.act) {                                                                                                                             \
        if (SIZES) {                                                                                                                         \
            const unsigned long long z = sweep_sizes_words<W>(s.pv);                                                                         \
            uint8_t *dst = S + (int64_t)s.start + s.o;                                                                                       \
            if (s.whole) {                                                                                                                   \
                __builtin_memcpy(dst, &z, N);                                                                                                \
            } else {                                                                                                                         \
                _Pragma("nounroll") for (int j = 0; j < N && s.o + j < s.len; ++j) dst[j] = (uint8_t)(z >> (8 * j));                         \
            }                                                                                                                                \
        } else {                                                                                                                             \
            uint8_t *dst = S + ((int64_t)s.start + s.o) * W;                                                                                 \
            if (s.whole) {                                                                                                                   \
                sweep_store(dst, s.pv);                                                                                                      \
            } else {                                                                                                                         \
                _Pragma("nounroll") for (int j = 0; j < N && s.o + j < s.len; ++j) {                                                         \
                    const unsigned long long e = W == 8 ? (j == 0 ? s.pv.a : s.pv.b) : ((j < N / 2 ? s.pv.a : s.pv.b) >> (8 * W * (j < N / 2 ? j : j - N / 2))); \
                    __builtin_memcpy(dst + (size_t)j * W, &e, W);                                                                            \
                }                                                                                                                            \
            }                                                                                                                                \
        }                                                                                                                                    \
    }
            HGX_SEGS(X)
#undef X
        }
    }
#undef HGX_SEGS
}
template <typename M, bool SUM> __device__ __forceinline__ int32_t track_size(M v) {
    return SUM ? (int32_t)v : (int32_t)__popcll((unsigned long long)v);
}
// size of the set at base i of a track whose words have 1 << wlog bytes
template <bool SUM> __device__ __forceinline__ int32_t track_size_at(const void *S, int64_t i, int wlog) {
    if (SUM)
        return ((const int32_t *)S)[i];
    switch (wlog) {
    case 0:
        return (int32_t)__popc((unsigned)((const uint8_t *)S)[i]);
    case 1:
        return (int32_t)__popc((unsigned)((const uint16_t *)S)[i]);
    case 2:
        return (int32_t)__popc(((const uint32_t *)S)[i]);
    default:
        return (int32_t)__popcll(((const unsigned long long *)S)[i]);
    }
}
// pS: the parent is the top of the scope — its A is the size of its own S (words of M; no separate pass); S (words of
// 1 << sLog bytes): the genome's own track, read where a segment has no parent — null: the constant ownSize (a genome without
// in-scope children).  AT: the type of a depth along the path — a byte when genome sets are counted (at most 64 genomes a
// group), which is a quarter of what the top-down sweep moves with 32-bit depths.
template <typename C, typename M, bool SUM, typename AT>
static __global__ void __launch_bounds__(256) k_sweep_down(const TopRec<C> *__restrict__ top, int64_t numTop, const BotRec<C> *__restrict__ pbot,
                                                           const AT *__restrict__ pA, const M *__restrict__ pS, const void *__restrict__ S, int sLog,
                                                           int32_t ownSize, AT *__restrict__ A, const int32_t *__restrict__ pEnc = nullptr) {
    constexpr int V = sizeof(AT) == 1 ? 8 : 4; // bases per lane and round: an 8- or 16-byte word of A
    struct AVec {
        AT e[V];
    };
    struct MVec {
        M e[V];
    };
    const int sub = (int)(threadIdx.x & 15);
    const int64_t groupsTotal = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; t < numTop; t += groupsTotal) {
        const TopRec<C> tr = top[t];
        const int64_t start = (int64_t)tr.start;
        const C len = (C)(top[t + 1].start - tr.start); // (offsets inside a segment in the tables' own width: k_sweep_up)
        // (pEnc — --noDupes: the parent's links to this genome: only the segment a parent's slot names goes up, mmapTopSegment.cpp:30-40;
        // the others' columns end with them, as an insertion's)
        const bool hasParent = tr.parentEnc >= 0 && (!pEnc || (int64_t)(pEnc[tr.parentEnc >> 1] >> 1) == t), rev = (tr.parentEnc & 1) != 0;
        const int64_t pstart = hasParent ? (int64_t)pbot[tr.parentEnc >> 1].start : 0;
        for (C o_ = (C)(sub * V); o_ < len; o_ += 16 * V) {
            const C o = o_ + V > len && len >= V ? len - V : o_; // (the last lane: the segment's last V bases, as in k_sweep_up)
            AVec a;
            if (o + V <= len && hasParent) {
                const int64_t pp = pstart + (rev ? len - o - V : o); // the V parent bases, in the parent's order
                if (pS) {
                    const MVec x = sweep_load<MVec>(pS + pp);
#pragma unroll
                    for (int j = 0; j < V; ++j)
                        a.e[j] = (AT)track_size<M, SUM>(x.e[rev ? V - 1 - j : j]);
                } else {
                    const AVec x = sweep_load<AVec>(pA + pp);
#pragma unroll
                    for (int j = 0; j < V; ++j)
                        a.e[j] = x.e[rev ? V - 1 - j : j];
                }
                sweep_store(A + start + o, a);
            } else { // (the end of a segment, or a segment without a parent — an insertion: its own set's size)
                for (int j = 0; j < V && o + j < len; ++j) {
                    const int64_t pp = pstart + (rev ? len - 1 - o - j : o + j);
                    A[start + o + j] = hasParent ? (pS ? (AT)track_size<M, SUM>(pS[pp]) : pA[pp])
                                                 : (AT)(S ? track_size_at<SUM>(S, start + o + j, sLog) : ownSize);
                }
            }
        }
    }
}
// k_sweep_down where depths are bytes and come from the parent's depths (every step of the path but the first): a copy of sixteen
// bases a lane as two 64-bit words, a segment in the other orientation by byte swaps (round 6: k_sweep_up_bytes has the why)
template <typename C>
static __global__ void __launch_bounds__(256) k_sweep_down_bytes(const TopRec<C> *__restrict__ top, int64_t numTop, const BotRec<C> *__restrict__ pbot,
                                                                 const uint8_t *__restrict__ pA, const void *__restrict__ S, int sLog, int32_t ownSize,
                                                                 uint8_t *__restrict__ A, const int32_t *__restrict__ pEnc) {
    const int sub = (int)(threadIdx.x & 7);
    const int64_t groupsTotal = ((int64_t)gridDim.x * blockDim.x) >> 3;
    for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; t < numTop; t += groupsTotal) {
        const TopRec<C> tr = top[t];
        const int64_t start = (int64_t)tr.start;
        const C len = (C)(top[t + 1].start - tr.start);
        const bool hasParent = tr.parentEnc >= 0 && (!pEnc || (int64_t)(pEnc[tr.parentEnc >> 1] >> 1) == t), rev = (tr.parentEnc & 1) != 0;
        const int64_t pstart = hasParent ? (int64_t)pbot[tr.parentEnc >> 1].start : 0;
        for (C o_ = (C)(sub * 16); o_ < len; o_ += 128) {
            const C o = o_ + 16 > len && len >= 16 ? len - 16 : o_;
            if (o + 16 <= len && hasParent) {
                sweep_store(A + start + o, sweep_bytes16<C>(pA + pstart, rev, len, o, true));
            } else { // (the end of a short segment, or a segment without a parent — an insertion: its own set's size)
#pragma nounroll
                for (int j = 0; j < 16 && o + j < len; ++j) {
                    const int64_t pp = pstart + (rev ? len - 1 - o - j : o + j);
                    A[start + o + j] = hasParent ? pA[pp] : (uint8_t)(S ? track_size_at<false>(S, start + o + j, sLog) : ownSize);
                }
            }
        }
    }
}
template <typename M, bool SUM, typename AT>
static __global__ void __launch_bounds__(256) k_sweep_top(const M *__restrict__ S, int64_t n, M own, AT *__restrict__ A) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        A[i] = (AT)track_size<M, SUM>(S ? S[i] : own);
}
// accumulate: more than 64 counted genomes go through the sweeps in groups of 64 (a group's genome sets are 64-bit words); the
// depth of a column is the sum of the groups' set sizes at the column's topmost ancestor
template <typename AT>
static __global__ void __launch_bounds__(256) k_sweep_out(const AT *__restrict__ A, int64_t first, int64_t count, int64_t step, int32_t sub,
                                                          int32_t *__restrict__ out, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (accumulate ? out[i] : 0) + (int32_t)A[first + i * step] - sub;
}


// k_sweep_out for byte depths and consecutive columns (step 1, a first pass: nothing to add to): sixteen columns a lane — one
// 16-byte load, four 16-byte stores (round 6: a column a lane was 370 us for the 224 M columns of config 5, 1.1 GB at 3 TB/s)
static __global__ void __launch_bounds__(256) k_sweep_out_bytes(const uint8_t *__restrict__ A, int64_t first, int64_t count, int32_t sub,
                                                                int32_t *__restrict__ out) {
    const int64_t chunks = (count + 15) >> 4;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = c << 4;
        if (i + 16 <= count) {
            const SweepW2 x = sweep_load<SweepW2>(A + first + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t w = (uint32_t)((q < 2 ? x.a : x.b) >> (32 * (q & 1)));
                struct {
                    int32_t x, y, z, w;
                } o;
                o.x = (int32_t)(w & 0xFFu) - sub;
                o.y = (int32_t)((w >> 8) & 0xFFu) - sub;
                o.z = (int32_t)((w >> 16) & 0xFFu) - sub;
                o.w = (int32_t)(w >> 24) - sub;
                sweep_store(out + i + 4 * q, o);
            }
        } else {
            for (int64_t j = i; j < count; ++j)
                out[j] = (int32_t)A[first + j] - sub;
        }
    }
}

// ---- halAlignmentDepth's lines on the device ----
// "%d\n" per column (alignmentDepth/halAlignmentDepth.cpp:246, 271, 305).  The host's threads counted and wrote a quarter of a
// billion lines in 40 ms beside a scan of 12; here a lane takes eight values, the lines' lengths cross the tiles by the one-pass
// scan (hgx_scan_kernels.hpp), the lane writes its lines where they belong.  ctl: {ticket, unused, total bytes (64 bits)}.
struct WigCtl {
    unsigned int ticket, _pad;
    unsigned long long bytes;
};
static constexpr uint32_t WIG_TILE = 2048;
static constexpr uint32_t WIG_STAGE = 8192; // bytes of LDS a tile's lines are staged in (2048 lines of up to three digits)
static __global__ void __launch_bounds__(256) k_wig_text(const int32_t *__restrict__ vals, uint32_t n, WigCtl *ctl, unsigned long long *tiles,
                                                         char *__restrict__ text) {
    const unsigned tile = lb_take_tile(&ctl->ticket);
    const unsigned numTiles = (n + WIG_TILE - 1) / WIG_TILE;
    if (tile >= numTiles)
        return;
    const uint32_t i0 = tile * WIG_TILE + threadIdx.x * 8;
    int32_t v[8];
    uint32_t len[8];
    unsigned long long w = 0;
    for (int j = 0; j < 8; ++j) {
        len[j] = 0;
        v[j] = 0;
        if (i0 + (uint32_t)j < n) {
            v[j] = vals[i0 + j];
            uint32_t u = v[j] < 0 ? 0u - (uint32_t)v[j] : (uint32_t)v[j];
            uint32_t k = v[j] < 0 ? 3 : 2; // a digit and the newline (and the sign)
            while (u >= 10) {
                u /= 10;
                ++k;
            }
            len[j] = k;
            w += k;
        }
    }
    const LbResult r = lb_scan_tile(tile, 0, w, tiles);
    // (round 6: a tile's lines are one stretch of the text — staged in LDS, a lane its own lines, and stored by all lanes sixteen
    // bytes each: a lane storing its twenty characters one by one was 0.65 ms for config 2's 55 M lines.  A tile of long numbers
    // that does not fit the stage goes straight to the text)
    __shared__ __attribute__((aligned(16))) char sText[WIG_STAGE];
    const bool staged = r.tileW <= (unsigned long long)WIG_STAGE;
    char *o = staged ? sText + (r.exW - r.baseW) : text + r.exW;
    for (int j = 0; j < 8; ++j) {
        if (!len[j])
            continue;
        uint32_t u = v[j] < 0 ? 0u - (uint32_t)v[j] : (uint32_t)v[j];
        if (v[j] < 0)
            *o++ = '-';
        const uint32_t digits = len[j] - (v[j] < 0 ? 2u : 1u);
        for (uint32_t d = digits; d > 0; --d) {
            o[d - 1] = (char)('0' + u % 10);
            u /= 10;
        }
        o += digits;
        *o++ = '\n';
    }
    if (staged) { // (the tile's own: every thread of the workgroup comes here)
        __syncthreads();
        char *dst = text + r.baseW;
        const uint32_t nbytes = (uint32_t)r.tileW;
        for (uint32_t at = threadIdx.x * 16u; at < nbytes; at += 256u * 16u) {
            if (at + 16u <= nbytes) {
                const SweepW2 x = *reinterpret_cast<const SweepW2 *>(sText + at);
                sweep_store(dst + at, x);
            } else {
                for (uint32_t i = at; i < nbytes; ++i)
                    dst[i] = sText[i];
            }
        }
    }
    if (tile == numTiles - 1 && threadIdx.x == 0)
        ctl->bytes = r.baseW + r.tileW;
}

} // namespace hgx
