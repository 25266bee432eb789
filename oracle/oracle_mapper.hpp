// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).
//
// Restatement of the reference's iterator algebra and pairwise block mapper:
//   api/impl/halSegmentIterator.cpp, halTopSegmentIterator.cpp,
//   halBottomSegmentIterator.cpp, halMappedSegment.cpp, halSegmentMapper.cpp.
// The control flow, containers (std::list, std::set of shared pointers with
// in-place clipping) and evaluation order follow the reference line by line so
// that set/merge semantics are reproduced, not re-derived.
#pragma once
#include "hal_oracle.hpp"
#include <algorithm>
#include <cassert>
#include <list>
#include <memory>
#include <set>

namespace orc {

// SegmentIterator state: api/inc/halSegmentIterator.h:115-117 (+ the segment's
// genome / array index held by the Segment object).
struct SegIt {
    const Alignment *al = nullptr;
    int g = -1;
    bool top = true;
    i64 idx = 0;
    i64 so = 0, eo = 0;
    bool rev = false;

    const Genome &G() const {
        return al->genomes[(size_t)g];
    }
    // api/mmap_impl/mmapTopSegment.h:78-80, mmapBottomSegment.h: length = next.start - start
    i64 segStart() const {
        return top ? G().tStart[(size_t)idx] : G().bStart[(size_t)idx];
    }
    i64 segLen() const {
        return top ? G().tStart[(size_t)idx + 1] - G().tStart[(size_t)idx] : G().bStart[(size_t)idx + 1] - G().bStart[(size_t)idx];
    }
    // api/impl/halSegmentIterator.cpp:38-40 + api/inc/halTopSegmentIterator.h:116
    i64 numSegs() const {
        return top ? G().numTop : G().numBot;
    }
    // halSegmentIterator.cpp:46-67
    i64 getStartPosition() const {
        return !rev ? segStart() + so : segStart() + segLen() - so - 1;
    }
    i64 getLength() const {
        return segLen() - eo - so;
    }
    i64 getEndPosition() const {
        return !rev ? getStartPosition() + (getLength() - 1) : getStartPosition() - (getLength() - 1);
    }
    // halSegmentIterator.cpp:86-108
    bool leftOf(i64 pos) const {
        return !rev ? getStartPosition() + getLength() <= pos : getStartPosition() < pos;
    }
    bool rightOf(i64 pos) const {
        return !rev ? getStartPosition() > pos : getStartPosition() - getLength() >= pos;
    }
    bool overlaps(i64 pos) const {
        return !leftOf(pos) && !rightOf(pos);
    }
    // halSegmentIterator.cpp:144-168
    void toReverse() {
        rev = !rev;
    }
    void toReverseInPlace() {
        rev = !rev;
        std::swap(so, eo);
    }
    void slice(i64 s, i64 e) {
        so = s;
        eo = e;
    }
    // halSegmentIterator.cpp:177-206
    void toLeft(i64 leftCutoff = NULL_INDEX) {
        if (!rev) {
            if (so == 0) {
                idx -= 1;
                eo = 0;
            } else {
                eo = segLen() - so;
                so = 0;
            }
            if (idx >= 0 && leftCutoff != NULL_INDEX && overlaps(leftCutoff))
                so = leftCutoff - segStart();
        } else {
            if (so == 0) {
                idx += 1;
                eo = 0;
            } else {
                eo = segLen() - so;
                so = 0;
            }
            if (idx < numSegs() && leftCutoff != NULL_INDEX && overlaps(leftCutoff))
                so = segStart() + segLen() - 1 - leftCutoff;
        }
    }
    // halSegmentIterator.cpp:208-238
    void toRight(i64 rightCutoff = NULL_INDEX) {
        if (!rev) {
            if (eo == 0) {
                idx += 1;
                so = 0;
            } else {
                so = segLen() - eo;
                eo = 0;
            }
            if (idx < numSegs() && rightCutoff != NULL_INDEX && overlaps(rightCutoff))
                eo = segStart() + segLen() - rightCutoff - 1;
        } else {
            if (eo == 0) {
                idx -= 1;
                so = 0;
            } else {
                so = segLen() - eo;
                eo = 0;
            }
            // note: the reference performs no lower-bound check here either
            if (idx >= 0 && rightCutoff != NULL_INDEX && overlaps(rightCutoff))
                eo = rightCutoff - segStart();
        }
    }
    // halSegmentIterator.cpp:240-299 (interpolation search; doubles only pick probes)
    void toSite(i64 position, bool doSlice = true) {
        i64 len = G().totalLength;
        i64 nseg = numSegs();
        double avgLen = (double)len / (double)nseg;
        i64 hint = (i64)std::min(nseg - 1., avgLen * ((double)position / (double)len));
        idx = hint;
        so = 0;
        eo = 0;
        if (position < 0) {
            idx = NULL_INDEX;
            return;
        } else if (position >= len) {
            idx = len;
            return;
        }
        i64 left = 0, leftStartPosition = 0, right = nseg - 1, rightStartPosition = len - 1;
        while (!overlaps(position)) {
            if (rightOf(position)) {
                right = idx;
                rightStartPosition = segStart();
                avgLen = double(rightStartPosition - leftStartPosition) / (right - left);
                i64 delta = (i64)std::max((rightStartPosition - position) / avgLen, 1.);
                delta = std::min(delta, idx);
                idx -= delta;
            } else {
                left = idx;
                leftStartPosition = segStart();
                avgLen = double(rightStartPosition - leftStartPosition) / (right - left);
                i64 delta = (i64)std::max((position - leftStartPosition) / avgLen, 1.);
                delta = std::min(delta, nseg - 1 - idx);
                idx += delta;
            }
        }
        if (doSlice) {
            so = position - segStart();
            eo = segStart() + segLen() - position - 1;
        }
    }

    // ---- top-segment predicates / links ----
    bool hasParent() const {
        return G().tParent[(size_t)idx] != NULL_INDEX;
    }
    bool hasNextParalogy() const {
        return G().tParalogy[(size_t)idx] != NULL_INDEX;
    }
    // api/mmap_impl/mmapTopSegment.cpp:30-40
    bool isCanonicalParalog() const {
        if (!hasParent())
            return false;
        const Genome &P = al->genomes[(size_t)G().parent];
        i64 slot = P.childSlotOf(g);
        return P.bChild[(size_t)slot][(size_t)G().tParent[(size_t)idx]] == idx;
    }
    bool hasChild(i64 slot) const {
        return G().bChild[(size_t)slot][(size_t)idx] != NULL_INDEX;
    }

    // api/impl/halBottomSegmentIterator.cpp:40-49
    void toParent(const SegIt &t) {
        al = t.al;
        g = t.G().parent;
        top = false;
        idx = t.G().tParent[(size_t)t.idx];
        so = t.so;
        eo = t.eo;
        rev = t.rev;
        if (t.G().tParentRev[(size_t)t.idx])
            toReverse();
    }
    // api/impl/halTopSegmentIterator.cpp:36-45
    void toChild(const SegIt &b, i64 slot) {
        al = b.al;
        g = b.G().children[(size_t)slot];
        top = true;
        idx = b.G().bChild[(size_t)slot][(size_t)b.idx];
        so = b.so;
        eo = b.eo;
        rev = b.rev;
        if (b.G().bChildRev[(size_t)slot][(size_t)b.idx])
            toReverse();
    }
    // api/impl/halTopSegmentIterator.cpp:55-81
    void toParseUp(const SegIt &b) {
        al = b.al;
        g = b.g;
        top = true;
        idx = b.G().bTopParse[(size_t)b.idx];
        rev = b.rev;
        i64 startPos = b.getStartPosition();
        while (startPos >= segStart() + segLen())
            ++idx;
        if (!rev) {
            so = startPos - segStart();
            i64 topEnd = segStart() + segLen();
            i64 botEnd = b.getStartPosition() + b.getLength();
            eo = std::max((i64)0, topEnd - botEnd);
        } else {
            so = segStart() + segLen() - 1 - startPos;
            i64 topEnd = segStart();
            i64 botEnd = b.getStartPosition() - b.getLength() + 1;
            eo = std::max((i64)0, botEnd - topEnd);
        }
    }
    // api/impl/halBottomSegmentIterator.cpp:51-76
    void toParseDown(const SegIt &t) {
        al = t.al;
        g = t.g;
        top = false;
        idx = t.G().tBotParse[(size_t)t.idx];
        rev = t.rev;
        i64 startPos = t.getStartPosition();
        while (startPos >= segStart() + segLen())
            ++idx;
        if (!rev) {
            so = startPos - segStart();
            i64 botEndSeg = segStart() + segLen();
            i64 topEnd = t.getStartPosition() + t.getLength();
            eo = std::max((i64)0, botEndSeg - topEnd);
        } else {
            so = segStart() + segLen() - 1 - startPos;
            i64 botStartSeg = segStart();
            i64 topEnd = t.getStartPosition() - t.getLength() + 1;
            eo = std::max((i64)0, topEnd - botStartSeg);
        }
    }
    // api/impl/halTopSegmentIterator.cpp:99-107
    void toNextParalogy() {
        bool r = G().tParentRev[(size_t)idx] != 0;
        idx = G().tParalogy[(size_t)idx];
        if ((G().tParentRev[(size_t)idx] != 0) != r)
            toReverse();
    }
    const Sequence *getSequence() const {
        return G().seqBySite(segStart());
    }
};

// api/inc/halMappedSegment.h:196-197 — a (source, target) pair of equal length.
struct MSeg {
    SegIt src, tgt;
    // SlicedSegment interface forwards to the target (halMappedSegment.cpp:298-311, 375-407)
    i64 getStartPosition() const {
        return tgt.getStartPosition();
    }
    i64 getEndPosition() const {
        return tgt.getEndPosition();
    }
    i64 getLength() const {
        return tgt.getLength();
    }
    bool getReversed() const {
        return tgt.rev;
    }
    i64 getStartOffset() const {
        return tgt.so;
    }
    i64 getEndOffset() const {
        return tgt.eo;
    }
    int getGenome() const {
        return tgt.g;
    }
    bool isTop() const {
        return tgt.top;
    }
    // halMappedSegment.cpp:395-402
    void slice(i64 startOffset, i64 endOffset) {
        i64 startDelta = startOffset - tgt.so;
        i64 endDelta = endOffset - tgt.eo;
        tgt.slice(startOffset, endOffset);
        src.slice(src.so + startDelta, src.eo + endDelta);
    }
};
typedef std::shared_ptr<MSeg> MSegPtr;

// halMappedSegment.cpp:254-277
inline int slowComp(const SegIt &s1, const SegIt &s2) {
    i64 sp1 = s1.getStartPosition(), ep1 = s1.getEndPosition();
    i64 sp2 = s2.getStartPosition(), ep2 = s2.getEndPosition();
    if (s1.rev)
        std::swap(sp1, ep1);
    if (s2.rev)
        std::swap(sp2, ep2);
    if (sp1 < sp2)
        return -1;
    if (sp1 > sp2)
        return 1;
    if (ep1 < ep2)
        return -1;
    if (ep1 > ep2)
        return 1;
    return 0;
}
// halMappedSegment.cpp:167-206.  The mixed top/bottom branch (boundComp, :208-252) only
// prunes before falling back to slowComp and the reference asserts res == slowComp, so the
// mixed case is restated as slowComp directly.
inline int fastComp(const SegIt &s1, const SegIt &s2) {
    if (s1.top != s2.top)
        return slowComp(s1, s2);
    if (s1.idx < s2.idx)
        return -1;
    if (s1.idx > s2.idx)
        return 1;
    i64 so1 = s1.so, eo1 = s1.eo;
    if (s1.rev)
        std::swap(so1, eo1);
    i64 so2 = s2.so, eo2 = s2.eo;
    if (s2.rev)
        std::swap(so2, eo2);
    if (so1 < so2)
        return -1;
    if (so1 > so2)
        return 1;
    if (eo1 > eo2)
        return -1;
    if (eo1 < eo2)
        return 1;
    return 0;
}
// halMappedSegment.cpp:36-61
inline bool lessThan(const MSeg &a, const MSeg &b) {
    int r = fastComp(a.tgt, b.tgt);
    if (r == 0)
        r = fastComp(a.src, b.src);
    return r == -1;
}
inline bool lessThanBySource(const MSeg &a, const MSeg &b) {
    int r = fastComp(a.src, b.src);
    if (r == 0)
        r = fastComp(a.tgt, b.tgt);
    return r == -1;
}
inline bool equalsM(const MSeg &a, const MSeg &b) {
    int r = fastComp(a.src, b.src);
    if (r == 0)
        r = fastComp(a.tgt, b.tgt);
    return r == 0;
}
struct MSegLess {
    bool operator()(const MSegPtr &a, const MSegPtr &b) const {
        return lessThan(*a, *b);
    }
};
typedef std::set<MSegPtr, MSegLess> MSegSet;
typedef std::list<MSegPtr> MSegList;

struct MapperStats { // "count mode" for the algorithmic-bytes figure
    u64 topDeref = 0, botDeref = 0;
};

// halMappedSegment.cpp:109-161
bool canMergeRightWith(const MSeg &self, const MSeg &next, const std::set<i64> *cutSet, const std::set<i64> *sourceCutSet);

// halSegmentMapper.cpp:639-670
size_t halMapSegment(const SegIt &source, MSegSet &out, int tgtGenome, const std::set<int> *genomesOnPath, bool doDupes,
                     i64 minLength, int coalescenceLimit, int mrca);

// api/impl/halCommon.cpp:123-152 / :176-187
int getLowestCommonAncestor(const Alignment &al, const std::set<int> &in);
void getGenomesInSpanningTree(const Alignment &al, const std::set<int> &in, std::set<int> &out);

} // namespace orc
