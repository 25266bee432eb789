// The column walk of a ColumnIterator with maxInsertLength > 0 (hal2maf --maxRefGap; api/impl/halColumnIterator.cpp:65-144,
// 246-405): between two columns of the reference the iterator walks the bases the reference lacks — ranges deleted on the
// way up (handleDeletion, :357-382) and inserted on the way down (handleInsertion, :384-405), found by Rearrangement's deletion
// and insertion cycles (api/impl/halRearrangement.cpp:133-176, 386-516) — as columns of their own, from a stack of ranges
// (api/inc/halColumnIteratorStack.h).  Which ranges are pushed, which columns are skipped (the visit cache, :749-819) and in
// what order they come is sequential state and is replayed on the host (hgx_columns_host.cpp); what a column CONTAINS and
// which indels its walk meets is a function of its first base alone, and is computed here for batches of columns:
//   * a column is asked for by (genome, position, reversed) — the stack entries lie in any genome, and a deleted range is
//     walked in the orientation its parent segment was reached in;
//   * every base the walk visits is reported in the reference's order, also the ones colMapInsert's filters keep out of the
//     column (noAncestors, targets): the visit cache looks at all of them (flag in the row);
//   * the indel the walk meets at a base is reported behind that base as an event: the range, its genome, its orientation.
// Same depth-first order as ColumnWalker (hgx_column_kernels.hpp); kept apart from it: the default path pays for none of this.
#pragma once
#include "hgx_column_kernels.hpp"

namespace hgx {

// index of the sequence that holds genome position pos (Genome::getSequenceBySite): seqStart[s] <= pos < seqStart[s + 1]
__device__ __forceinline__ int gap_seq_of(const int64_t *__restrict__ seqStart, int numSeq, int64_t pos) {
    int lo = 0, hi = numSeq;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seqStart[mid] <= pos)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

struct GapAsk { // a column: the base its walk starts from (ColumnIteratorStack::Entry's _sequence / _index / _reversed)
    int64_t pos;
    int32_t genome;
    int32_t reversed;
};

// row kinds (ColumnRow::_pad[0]); _pad[1] = the level of the walk's upward chain the row belongs to (0: the column's own
// genome; a parent base reached by updateParent: its level, 1 for the first; events of handleDeletion at a parse-up: the level
// of the parent base they hang on) — the host needs it to know which deletions of an abandoned column were still made
enum : uint8_t { GAP_ROW = 0, GAP_ROW_HIDDEN = 1, GAP_DELETION = 2, GAP_INSERTION = 3, GAP_EVENT_END = 4, GAP_ROW_UP = 8 /* flag: reached by updateParent */,
                 GAP_ROW_RING = 16 /* flag: a paralog updateNextTopDup's loop inserts */ };

// a whole segment seen through an iterator (what the atomic, gap-threshold-0 gapped iterators of Rearrangement degenerate to:
// halGappedTopSegmentIterator.cpp / halGappedBottomSegmentIterator.cpp with _atomic and _gapThreshold == 0)
struct GapSeg {
    int32_t g, idx;
    bool top, rev;
};

template <typename C> struct GapTables {
    const GenomeDesc *desc;
    __device__ __forceinline__ const TopRec<C> *top(int g) const { return (const TopRec<C> *)desc[g].top; }
    __device__ __forceinline__ const BotRec<C> *bot(int g) const { return (const BotRec<C> *)desc[g].bot; }
    __device__ __forceinline__ int64_t segStart(const GapSeg &s) const {
        return s.top ? (int64_t)top(s.g)[s.idx].start : (int64_t)bot(s.g)[s.idx].start;
    }
    __device__ __forceinline__ int64_t segEnd(const GapSeg &s) const { // last base
        return (s.top ? (int64_t)top(s.g)[s.idx + 1].start : (int64_t)bot(s.g)[s.idx + 1].start) - 1;
    }
    __device__ __forceinline__ int sequenceOf(const GapSeg &s) const {
        return gap_seq_of(desc[s.g].seqStart, desc[s.g].numSeq, segStart(s));
    }
    // Segment::isFirst / isLast (api/mmap_impl/mmapTopSegment.h:102-111): first / last segment of its sequence
    __device__ __forceinline__ bool segFirst(const GapSeg &s) const {
        return s.idx == 0 || segStart(s) == desc[s.g].seqStart[sequenceOf(s)];
    }
    __device__ __forceinline__ bool segLast(const GapSeg &s) const {
        return segEnd(s) + 1 == desc[s.g].seqStart[sequenceOf(s) + 1];
    }
    // SegmentIterator::isFirst / isLast (halSegmentIterator.cpp:110-116)
    __device__ __forceinline__ bool isFirst(const GapSeg &s) const { return !s.rev ? segFirst(s) : segLast(s); }
    __device__ __forceinline__ bool isLast(const GapSeg &s) const { return !s.rev ? segLast(s) : segFirst(s); }
    __device__ __forceinline__ bool hasParent(const GapSeg &s) const { return top(s.g)[s.idx].parentEnc >= 0; }
    __device__ __forceinline__ GapSeg toParent(const GapSeg &t) const { // halBottomSegmentIterator.cpp:40-49
        const int32_t enc = top(t.g)[t.idx].parentEnc;
        return GapSeg{desc[t.g].parent, enc >> 1, false, (bool)(t.rev ^ ((enc & 1) != 0))};
    }
    __device__ __forceinline__ GapSeg toRight(GapSeg s) const { // a whole segment's step (halSegmentIterator.cpp:208-238)
        s.idx += s.rev ? -1 : 1;
        return s;
    }
    __device__ __forceinline__ GapSeg toLeft(GapSeg s) const {
        s.idx += s.rev ? 1 : -1;
        return s;
    }
    // GappedBottomSegmentIterator::adjacentTo (halGappedBottomSegmentIterator.cpp:293-337) for one segment each: `other` is the
    // segment next to this one on either side
    __device__ __forceinline__ bool adjacentTo(const GapSeg &self, const GapSeg &other) const {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 0 ? !isFirst(self) : !isLast(self)) {
                if (!isFirst(other) && toLeft(other).idx == self.idx)
                    return true;
                if (!isLast(other) && toRight(other).idx == self.idx)
                    return true;
            }
        }
        return false;
    }
    // Rearrangement::identifyDeletionFromLeftBreakpoint (halRearrangement.cpp:133-140) = scanDeletionCycle (:458-516) and no
    // child under the candidate; true: [first, last] = the deleted range (getDeletedRange, :142-154), in the parent genome
    __device__ bool deletion(const GapSeg &cur, int64_t &first, int64_t &last) const {
        const bool isF = isFirst(cur), isL = isLast(cur);
        if (!hasParent(cur) || (isF && isL))
            return false;
        GapSeg leftParent = toParent(cur);
        bool found = false;
        if (isL) {
            if (!isFirst(leftParent)) {
                leftParent = toLeft(leftParent);
                found = true;
            } else if (!isLast(leftParent)) {
                leftParent = toRight(leftParent);
                found = true;
            }
        } else {
            const GapSeg right = toRight(cur);
            if (!hasParent(right))
                return false;
            GapSeg rightParent = toParent(right);
            if (sequenceOf(leftParent) == sequenceOf(rightParent)) {
                leftParent.rev = false; // "don't care about inversions": both made forward, left of right
                rightParent.rev = false;
                if (rightParent.idx < leftParent.idx) {
                    const GapSeg t = leftParent;
                    leftParent = rightParent;
                    rightParent = t;
                }
                if (isLast(leftParent))
                    return false;
                leftParent = toRight(leftParent);
                found = adjacentTo(leftParent, rightParent);
            }
        }
        if (!found)
            return false;
        const int slot = desc[cur.g].slotInParent; // the gapped bottom iterator's child index (toParent)
        if (desc[leftParent.g].child[slot][leftParent.idx] >= 0)
            return false; // hasChild: a transposition's source, not a deletion
        first = segStart(leftParent);
        last = segEnd(leftParent);
        return true;
    }
    // Rearrangement::identifyInsertionFromLeftBreakpoint (:156-163) = scanInsertionCycle (:386-456) and no parent over the
    // candidate; [first, last] as getInsertedRange computes it (:165-176: from the iterator's start position upwards — for a
    // reversed iterator that is its high end, so the range lies behind the segment; kept as it is)
    __device__ bool insertion(const GapSeg &cur, int64_t &first, int64_t &last) const {
        GapSeg next = cur, right = cur, left = cur;
        // adjacent insertions are eaten so that they are not counted twice
        while (!hasParent(next) && !isLast(next)) {
            right = toRight(next);
            if (!hasParent(right))
                next = right;
            else
                break;
        }
        right = next;
        const bool isF = isFirst(cur), isL = isLast(right);
        if (isF && isL)
            return false;
        bool found = false;
        if (isF) {
            right = toRight(right);
            if (!hasParent(cur))
                found = true;
            else if (hasParent(right))
                found = !adjacentTo(toParent(right), toParent(cur));
        } else if (isL) {
            left = toLeft(left);
            if (!hasParent(cur))
                found = true;
            else if (hasParent(left))
                found = !adjacentTo(toParent(left), toParent(cur));
        } else {
            left = toLeft(left);
            right = toRight(right);
            if (hasParent(left) && hasParent(right)) {
                const GapSeg lp = toParent(left), rp = toParent(right);
                if (adjacentTo(lp, rp))
                    found = true;
                else if (isFirst(lp) || isLast(lp) || isFirst(rp) || isLast(rp))
                    found = sequenceOf(lp) == sequenceOf(rp);
            }
        }
        if (!found || hasParent(cur))
            return false;
        first = !cur.rev ? segStart(cur) : segEnd(cur);
        last = first + (segEnd(cur) - segStart(cur));
        return true;
    }
};

enum : uint32_t { FR_DELETION = 5 };

// V: visitor with  void row(int genome, int64_t pos, bool rev, bool reported, int depth, bool up, bool ring)
//                  void event(uint8_t kind, int level, int genome, int64_t first, int64_t last, bool reversed)
// depth: how deep in the recursion the call that inserts the base sits — 0 the column's own base, one more for the parent base an
// updateParent inserts (on the upward chain depth and level are the same number), for the child bases of updateChild, and for the
// paralogs updateNextTopDup goes round (ring: they hang one deeper than the segment whose ring it is, their own children another
// one deeper).  The host needs it where a walk is abandoned at a base seen before: the paralogy loop of updateNextTopDup does not
// look at _break (halColumnIterator.cpp:653-680), so the rest of a ring whose member's subtree the walk was abandoned in is still
// inserted — those bases, and only those, are the ring rows at the depth of a ring member the abandoned base lies under.
template <typename C> struct GapWalker {
    const ColumnParams &P;
    GapTables<C> tab;
    Frame stack[COL_STACK];
    int sp = 0;
    bool overflow = false;
    __device__ GapWalker(const ColumnParams &p) : P(p) { tab.desc = p.desc; }
    // depth: of the base the frame belongs to (what it inserts lies one deeper)
    __device__ __forceinline__ void push(uint32_t kind, int g, int32_t idx, int32_t so, bool rev, int32_t extra, int depth) {
        if (sp >= COL_STACK || depth >= 254 || g >= 65536) {
            overflow = true;
            return;
        }
        Frame f;
        f.idx = idx;
        f.so = so;
        f.extra = extra;
        f.meta = kind | ((uint32_t)rev << 3) | ((uint32_t)g << 4) | ((uint32_t)depth << 20);
        stack[sp++] = f;
    }
    template <typename REC> __device__ __forceinline__ int64_t posOf(const REC *segs, int32_t idx, int32_t so, bool rev) const {
        return !rev ? (int64_t)segs[idx].start + so : (int64_t)segs[idx + 1].start - 1 - so;
    }
    template <typename V> __device__ __forceinline__ void insert(V &visit, int g, int64_t pos, bool rev, int depth, bool up, bool ring) const {
        visit.row(g, pos, rev, (!P.noAncestors || P.desc[g].numChildren == 0) && bit(P.targetMask, g), depth, up, ring);
    }
    // handleDeletion (halColumnIterator.cpp:357-382) at the base with offset so (iteration order) of top segment t of genome g
    template <typename V> __device__ __forceinline__ void handleDeletion(V &visit, int g, int32_t t, int32_t so, bool rev, int level) const {
        if (P.noGapEvents)
            return; // (maxInsertLength == 0: the handlers do nothing, :358, :385)
        const TopRec<C> *T = tab.top(g);
        if (T[t].parentEnc < 0 || (int64_t)so != (int64_t)T[t + 1].start - (int64_t)T[t].start - 1)
            return; // no parent, or not immediately left of the breakpoint (end offset 0)
        int64_t first, last;
        const GapSeg cur{g, t, true, rev};
        if (tab.deletion(cur, first, last))
            visit.event(GAP_DELETION, level, P.desc[g].parent, first, last, tab.toParent(cur).rev);
    }
    // handleInsertion (:384-405)
    template <typename V> __device__ __forceinline__ void handleInsertion(V &visit, int g, int32_t t, int32_t so, bool rev, int level) const {
        if (P.noGapEvents)
            return;
        const TopRec<C> *T = tab.top(g);
        if (T[t].parentEnc < 0 || (int64_t)so != (int64_t)T[t + 1].start - (int64_t)T[t].start - 1)
            return;
        const GapSeg in{g, t, true, rev};
        if (tab.isLast(in))
            return;
        int64_t first, last;
        if (tab.insertion(tab.toRight(in), first, last))
            visit.event(GAP_INSERTION, level, g, first, last, rev);
    }
    // the walk of the column of base p of genome R, the iterator reversed or not (recursiveUpdate, :246-355)
    template <typename V> __device__ void run(int R, int64_t p, bool rrev, V &visit) {
        const GenomeDesc &RD = P.desc[R];
        sp = 0;
        if (RD.numTop > 0) {
            const TopRec<C> *T = tab.top(R);
            int64_t lo = 0, hi = RD.numTop;
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)T[mid].start <= p)
                    lo = mid;
                else
                    hi = mid;
            }
            const int32_t t = (int32_t)lo;
            const TopRec<C> tr = T[t];
            const int32_t so = !rrev ? (int32_t)(p - (int64_t)tr.start) : (int32_t)((int64_t)T[t + 1].start - 1 - p);
            insert(visit, R, p, rrev, 0, false, false);
            handleDeletion(visit, R, t, so, rrev, 0);
            if (tr.botParse >= 0)
                push(FR_PARSEDOWN, R, t, so, rrev, 0, 0);
            if (!P.onlyOrthologs && tr.paralogy >= 0)
                push(FR_RING, R, t, so, rrev, t, 0);
            if (tr.parentEnc >= 0)
                push(FR_UP, R, t, so, rrev, 1, 0);
        } else {
            const BotRec<C> *B = tab.bot(R);
            int64_t lo = 0, hi = RD.numBot;
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)B[mid].start <= p)
                    lo = mid;
                else
                    hi = mid;
            }
            const int32_t b = (int32_t)lo;
            const int32_t so = !rrev ? (int32_t)(p - (int64_t)B[b].start) : (int32_t)((int64_t)B[b + 1].start - 1 - p);
            insert(visit, R, p, rrev, 0, false, false);
            for (int i = RD.numChildren - 1; i >= 0; --i)
                if (RD.child[i][b] >= 0)
                    push(FR_CHILD, R, b, so, rrev, i, 0);
        }
        while (sp > 0) {
            const Frame f = stack[--sp];
            const uint32_t kind = f.meta & 7u;
            const bool rev = (f.meta >> 3) & 1u;
            const int g = (int)((f.meta >> 4) & 0xFFFFu);
            const int depth = (int)(f.meta >> 20);
            const GenomeDesc &D = P.desc[g];
            if (kind == FR_UP) { // updateParent (:556-605); f.extra: the level of the parent base
                const TopRec<C> tr = tab.top(g)[f.idx];
                if (tr.parentEnc >= 0 && D.parent >= 0 && bit(P.scopeMask, D.parent)) {
                    const int pg = D.parent;
                    const GenomeDesc &PD = P.desc[pg];
                    const int32_t b = tr.parentEnc >> 1;
                    if (!P.noDupes || (PD.child[D.slotInParent][b] >> 1) == f.idx) {
                        const bool brev = rev ^ ((tr.parentEnc & 1) != 0);
                        insert(visit, pg, posOf(tab.bot(pg), b, f.so, brev), brev, depth + 1, true, false); // (depth + 1 == f.extra, the level)
                        for (int i = PD.numChildren - 1; i >= 0; --i) // siblings: after the parse-up branch and its deletion
                            if (i != D.slotInParent && PD.child[i][b] >= 0)
                                push(FR_CHILD, pg, b, f.so, brev, i, depth + 1);
                        if (PD.parent >= 0) {
                            push(FR_DELETION, pg, b, f.so, brev, f.extra, depth + 1); // handleDeletion(parent's top parse), :587-589
                            push(FR_PARSEUP, pg, b, f.so, brev, f.extra, depth + 1);
                        }
                    }
                }
            } else if (kind == FR_PARSEUP || kind == FR_DELETION) {
                // updateParseUp (:683-709) / the deletion check on the same top segment once that branch is done
                const BotRec<C> *B = tab.bot(g);
                const int32_t tp = B[f.idx].topParse;
                if (tp >= 0) {
                    const int64_t pos = posOf(B, f.idx, f.so, rev);
                    const TopRec<C> *T = tab.top(g);
                    int32_t j = tp;
                    while ((int64_t)T[j + 1].start <= pos)
                        ++j;
                    const TopRec<C> tj = T[j];
                    const int32_t so = !rev ? (int32_t)(pos - (int64_t)tj.start) : (int32_t)((int64_t)T[j + 1].start - 1 - pos);
                    if (kind == FR_DELETION) {
                        handleDeletion(visit, g, j, so, rev, f.extra);
                    } else {
                        if (!P.onlyOrthologs && tj.paralogy >= 0)
                            push(FR_RING, g, j, so, rev, j, depth);
                        if (tj.parentEnc >= 0)
                            push(FR_UP, g, j, so, rev, f.extra + 1, depth);
                    }
                }
            } else if (kind == FR_CHILD) { // updateChild (:607-640)
                const int slot = f.extra;
                const int32_t enc = D.child[slot][f.idx];
                const int cg = D.childGenome[slot];
                if (enc >= 0 && bit(P.scopeMask, cg)) {
                    const int32_t t = enc >> 1;
                    const bool crev = rev ^ ((enc & 1) != 0);
                    const TopRec<C> ct = tab.top(cg)[t];
                    insert(visit, cg, !crev ? (int64_t)ct.start + f.so : (int64_t)tab.top(cg)[t + 1].start - 1 - f.so, crev, depth + 1, false, false);
                    handleInsertion(visit, cg, t, f.so, crev, 0);
                    if (ct.botParse >= 0)
                        push(FR_PARSEDOWN, cg, t, f.so, crev, 0, depth + 1);
                    if (ct.paralogy >= 0)
                        push(FR_RING, cg, t, f.so, crev, t, depth + 1);
                }
            } else if (kind == FR_RING) { // updateNextTopDup (:642-681), one ring member per frame
                const TopRec<C> *T = tab.top(g);
                const TopRec<C> cur = T[f.idx];
                const int32_t first = f.extra;
                const bool startOfRing = f.idx == first;
                bool go = !P.noDupes && cur.paralogy >= 0 && D.parent >= 0 && bit(P.scopeMask, D.parent);
                if (go && !startOfRing)
                    go = cur.paralogy != first;
                if (go) {
                    const int32_t nxt = cur.paralogy;
                    const TopRec<C> nr = T[nxt];
                    const bool nrev = rev ^ ((nr.parentEnc & 1) != (cur.parentEnc & 1));
                    insert(visit, g, posOf(T, nxt, f.so, nrev), nrev, depth + 1, false, true);
                    handleInsertion(visit, g, nxt, f.so, nrev, 0);
                    if (nr.paralogy >= 0 && nr.paralogy != first)
                        push(FR_RING, g, nxt, f.so, nrev, first, depth); // (the rest of the ring: the same loop)
                    if (nr.botParse >= 0)
                        push(FR_PARSEDOWN, g, nxt, f.so, nrev, 0, depth + 1);
                }
            } else { // FR_PARSEDOWN: updateParseDown (:711-744)
                const TopRec<C> *T = tab.top(g);
                const int32_t bp = T[f.idx].botParse;
                if (bp >= 0) {
                    const int64_t pos = posOf(T, f.idx, f.so, rev);
                    const BotRec<C> *B = tab.bot(g);
                    int32_t j = bp;
                    while ((int64_t)B[j + 1].start <= pos)
                        ++j;
                    const int32_t so = !rev ? (int32_t)(pos - (int64_t)B[j].start) : (int32_t)((int64_t)B[j + 1].start - 1 - pos);
                    for (int i = D.numChildren - 1; i >= 0; --i)
                        if (D.child[i][j] >= 0)
                            push(FR_CHILD, g, j, so, rev, i, depth);
                }
            }
        }
    }
};

struct GapCountVisitor {
    uint32_t n = 0;
    __device__ __forceinline__ void row(int, int64_t, bool, bool, int, bool, bool) { ++n; }
    __device__ __forceinline__ void event(uint8_t, int, int, int64_t, int64_t, bool) { n += 2; }
};
struct GapRowVisitor {
    ColumnRow *dst;
    const GenomeDesc *desc;
    uint32_t n = 0;
    __device__ __forceinline__ void row(int g, int64_t pos, bool rev, bool reported, int depth, bool up, bool ring) {
        ColumnRow r;
        r.pos = pos;
        r.genome = g;
        r.rev = rev;
        char c = 'N';
        const uint8_t *dna = desc[g].dna;
        if (dna) {
            const uint8_t b = dna[pos >> 1];
            c = "acgtn\0\0\0ACGTN\0\0"[(pos & 1) ? (b & 0x0F) : (b >> 4)]; // dnaUnpack, halCommon.h:187-190
            if (rev) {
                switch (c) { // reverseComplement (halCommon.h:45-75)
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
        r._pad[0] = (uint8_t)((reported ? GAP_ROW : GAP_ROW_HIDDEN) | (up ? GAP_ROW_UP : 0) | (ring ? GAP_ROW_RING : 0));
        r._pad[1] = (uint8_t)depth; // (on the upward chain: the level)
        dst[n++] = r;
    }
    __device__ __forceinline__ void event(uint8_t kind, int level, int g, int64_t first, int64_t last, bool reversed) {
        ColumnRow r;
        r.pos = first;
        r.genome = g;
        r.rev = reversed;
        r.base = 0;
        r._pad[0] = kind;
        r._pad[1] = (uint8_t)level;
        dst[n++] = r;
        r.pos = last;
        r._pad[0] = GAP_EVENT_END;
        dst[n++] = r;
    }
};

// pass 1: rows (and event half-rows) per asked column; pass 2: the rows at the scanned offsets
template <typename C>
__global__ void __launch_bounds__(256) k_gap_count(ColumnParams P, const GapAsk *__restrict__ asks, uint32_t *__restrict__ counts) {
    GapWalker<C> w(P);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.count; i += (int64_t)gridDim.x * blockDim.x) {
        GapCountVisitor v;
        const GapAsk a = asks[i];
        w.run(a.genome, a.pos, a.reversed != 0, v);
        counts[i] = v.n;
    }
    if (w.overflow)
        *P.error = 1;
}
template <typename C>
__global__ void __launch_bounds__(256) k_gap_rows(ColumnParams P, const GapAsk *__restrict__ asks, const uint64_t *__restrict__ rowOffset,
                                                  ColumnRow *__restrict__ rows) {
    GapWalker<C> w(P);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.count; i += (int64_t)gridDim.x * blockDim.x) {
        GapRowVisitor v;
        v.dst = rows + rowOffset[i];
        v.desc = P.desc;
        const GapAsk a = asks[i];
        w.run(a.genome, a.pos, a.reversed != 0, v);
    }
    if (w.overflow)
        *P.error = 1;
}

} // namespace hgx
