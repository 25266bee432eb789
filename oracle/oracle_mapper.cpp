// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).
// Restatement of api/impl/halSegmentMapper.cpp and the tree helpers of api/impl/halCommon.cpp.
#include "oracle_mapper.hpp"
#include <map>

namespace orc {

// ---------------------------------------------------------------------------------------------
// api/impl/halCommon.cpp:108-152 getLowestCommonAncestor
static size_t lcaRecursive(const Alignment &al, int genome, const std::set<int> &in, std::map<int, size_t> &table) {
    size_t score = in.count(genome) ? 1 : 0;
    for (int c : al.genomes[(size_t)genome].children)
        score += lcaRecursive(al, c, in, table);
    table[genome] = score;
    return score;
}
int getLowestCommonAncestor(const Alignment &al, const std::set<int> &in) {
    if (in.empty())
        return -1;
    std::map<int, size_t> table;
    int lca = al.root();
    lcaRecursive(al, lca, in, table);
    bool found = false;
    while (!found) {
        found = true;
        size_t score = table[lca];
        const std::vector<int> &childs = al.genomes[(size_t)lca].children;
        for (size_t i = 0; found && i < childs.size(); ++i) {
            if (table[childs[i]] == score) {
                lca = childs[i];
                found = false;
            }
        }
    }
    return lca;
}
// api/impl/halCommon.cpp:156-187 getGenomesInSpanningTree
static bool spanningRecursive(const Alignment &al, int genome, std::set<int> &out, bool below = false) {
    bool above = false;
    if (out.count(genome)) {
        below = true;
        above = true;
    }
    for (int c : al.genomes[(size_t)genome].children) {
        bool childAbove = spanningRecursive(al, c, out, below);
        above = above || childAbove;
    }
    if (above && below)
        out.insert(genome);
    return above;
}
void getGenomesInSpanningTree(const Alignment &al, const std::set<int> &in, std::set<int> &out) {
    int lca = getLowestCommonAncestor(al, in);
    if (lca < 0)
        return;
    out = in;
    out.insert(lca);
    spanningRecursive(al, lca, out);
}

// ---------------------------------------------------------------------------------------------
// api/impl/halMappedSegment.cpp:109-161
bool canMergeRightWith(const MSeg &self, const MSeg &next, const std::set<i64> *cutSet, const std::set<i64> *sourceCutSet) {
    bool ret = false;
    const SegIt &ref = self.src;
    const SegIt &nextRef = next.src;
    i64 sourceCut = 0, cut = 0;
    if (self.getReversed() == next.getReversed() && ref.rev == nextRef.rev) {
        i64 qdelta, rdelta;
        if (!self.getReversed() && !ref.rev) {
            qdelta = next.getStartPosition() - self.getEndPosition();
            rdelta = nextRef.getStartPosition() - ref.getEndPosition();
            cut = self.getEndPosition();
            sourceCut = ref.getEndPosition();
        } else if (self.getReversed() && ref.rev) {
            qdelta = next.getEndPosition() - self.getStartPosition();
            rdelta = nextRef.getEndPosition() - ref.getStartPosition();
            cut = self.getStartPosition();
            sourceCut = ref.getStartPosition();
        } else if (!self.getReversed() && ref.rev) {
            qdelta = next.getStartPosition() - self.getEndPosition();
            rdelta = ref.getEndPosition() - nextRef.getStartPosition();
            cut = self.getEndPosition();
            sourceCut = nextRef.getStartPosition();
        } else {
            qdelta = next.getEndPosition() - self.getStartPosition();
            rdelta = ref.getStartPosition() - nextRef.getEndPosition();
            cut = self.getStartPosition();
            sourceCut = nextRef.getEndPosition();
        }
        ret = qdelta == 1 && rdelta == 1;
    }
    if (ret) {
        if (sourceCutSet != nullptr && sourceCutSet->count(sourceCut))
            ret = false;
        else if (cutSet != nullptr && cutSet->count(cut))
            ret = false;
    }
    return ret;
}

// ---------------------------------------------------------------------------------------------
static size_t mapSelf(MSegPtr mappedSeg, MSegList &results, i64 minLength);

static MSegPtr newMSeg(const SegIt &s, const SegIt &t) {
    MSegPtr p(new MSeg);
    p->src = s;
    p->tgt = t;
    return p;
}

// halSegmentMapper.cpp:25-80
static size_t mapUp(MSegPtr mappedSeg, MSegList &results, bool doDupes, i64 minLength) {
    size_t added = 0;
    if (mappedSeg->isTop()) {
        const SegIt &topSegIt = mappedSeg->tgt;
        if (topSegIt.hasParent() && topSegIt.getLength() >= minLength && (doDupes || topSegIt.isCanonicalParalog())) {
            SegIt botSegIt;
            botSegIt.toParent(topSegIt);
            mappedSeg->tgt = botSegIt;
            results.push_back(mappedSeg);
            ++added;
        }
    } else {
        i64 rightCutoff = mappedSeg->getEndPosition();
        const SegIt botSegIt = mappedSeg->tgt;
        i64 startOffset = botSegIt.so, endOffset = botSegIt.eo;
        SegIt topSegIt;
        topSegIt.toParseUp(botSegIt);
        do {
            SegIt newTopSegIt = topSegIt;
            // map the new target back to see how the offsets changed; apply the deltas to the source
            SegIt backBotSegIt = botSegIt;
            backBotSegIt.toParseDown(newTopSegIt);
            i64 startBack = backBotSegIt.so, endBack = backBotSegIt.eo;
            SegIt newSourceSegIt = mappedSeg->src;
            i64 startDelta = startBack - startOffset, endDelta = endBack - endOffset;
            newSourceSegIt.slice(newSourceSegIt.so + startDelta, newSourceSegIt.eo + endDelta);
            MSegPtr newMappedSeg = newMSeg(newSourceSegIt, newTopSegIt);
            added += mapUp(newMappedSeg, results, doDupes, minLength);
            if (topSegIt.getEndPosition() != rightCutoff)
                topSegIt.toRight(rightCutoff);
            else
                break;
        } while (true);
    }
    return added;
}

struct LessSourcePtr {
    bool operator()(const MSegPtr &a, const MSegPtr &b) const {
        return lessThanBySource(*a, *b);
    }
};
struct EqualToPtr {
    bool operator()(const MSegPtr &a, const MSegPtr &b) const {
        return equalsM(*a, *b);
    }
};

// halSegmentMapper.cpp:85-125
static size_t mapRecursiveUp(MSegList &input, MSegList &results, int tgtGenome, i64 minLength, const Alignment &al) {
    MSegList *inputPtr = &input;
    MSegList *outputPtr = &results;
    if (inputPtr->empty() || (*inputPtr->begin())->getGenome() == tgtGenome) {
        results = *inputPtr;
        return 0;
    }
    int curGenome = (*inputPtr->begin())->getGenome();
    int nextGenome = al.genomes[(size_t)curGenome].parent;
    if (nextGenome < 0)
        throw std::runtime_error("Reached top of tree when attempting to recursively map up");
    for (auto i = inputPtr->begin(); i != inputPtr->end(); ++i)
        mapUp(*i, *outputPtr, true, minLength);
    if (nextGenome != tgtGenome) {
        std::swap(inputPtr, outputPtr);
        outputPtr->clear();
        mapRecursiveUp(*inputPtr, *outputPtr, tgtGenome, minLength, al);
    }
    if (outputPtr != &results)
        results = *outputPtr;
    results.sort(LessSourcePtr());
    results.unique(EqualToPtr());
    return results.size();
}

// halSegmentMapper.cpp:128-186
static size_t mapDown(MSegPtr mappedSeg, i64 childIndex, MSegList &results, i64 minLength) {
    size_t added = 0;
    if (!mappedSeg->isTop()) {
        const SegIt &botSegIt = mappedSeg->tgt;
        if (botSegIt.hasChild(childIndex) && botSegIt.getLength() >= minLength) {
            SegIt topSegIt;
            topSegIt.toChild(botSegIt, childIndex);
            mappedSeg->tgt = topSegIt;
            results.push_back(mappedSeg);
            ++added;
        }
    } else {
        i64 rightCutoff = mappedSeg->getEndPosition();
        const SegIt topSegIt = mappedSeg->tgt;
        i64 startOffset = topSegIt.so, endOffset = topSegIt.eo;
        SegIt botSegIt;
        botSegIt.toParseDown(topSegIt);
        do {
            SegIt newBotSegIt = botSegIt;
            SegIt backTopSegIt = topSegIt;
            backTopSegIt.toParseUp(newBotSegIt);
            i64 startBack = backTopSegIt.so, endBack = backTopSegIt.eo;
            SegIt newSourceSegIt = mappedSeg->src;
            i64 startDelta = startBack - startOffset, endDelta = endBack - endOffset;
            newSourceSegIt.slice(newSourceSegIt.so + startDelta, newSourceSegIt.eo + endDelta);
            MSegPtr newMappedSeg = newMSeg(newSourceSegIt, newBotSegIt);
            added += mapDown(newMappedSeg, childIndex, results, minLength);
            if (botSegIt.getEndPosition() != rightCutoff)
                botSegIt.toRight(rightCutoff);
            else
                break;
        } while (true);
    }
    return added;
}

// halSegmentMapper.cpp:191-260
static size_t mapRecursiveDown(MSegList &input, MSegList &results, int tgtGenome, const std::set<int> &onPath, bool doDupes,
                               i64 minLength, const Alignment &al) {
    MSegList *inputPtr = &input;
    MSegList *outputPtr = &results;
    if (inputPtr->empty()) {
        results = *inputPtr;
        return 0;
    }
    int curGenome = (*inputPtr->begin())->getGenome();
    if (curGenome == tgtGenome) {
        results = *inputPtr;
        return 0;
    }
    // find the child on the path: first child that is the target or on the path (:208-219; the
    // reference compares names, genome names are unique so indices are equivalent)
    int nextGenome = -1;
    i64 nextChildIndex = -1;
    const std::vector<int> &childs = al.genomes[(size_t)curGenome].children;
    for (size_t child = 0; nextGenome < 0 && child < childs.size(); ++child) {
        if (childs[child] == tgtGenome || onPath.count(childs[child])) {
            nextGenome = childs[child];
            nextChildIndex = (i64)child;
        }
    }
    if (nextGenome < 0)
        throw std::runtime_error("Could not find correct child that leads to target");
    for (auto i = inputPtr->begin(); i != inputPtr->end(); ++i)
        mapDown(*i, nextChildIndex, *outputPtr, minLength);
    if (doDupes) {
        std::swap(inputPtr, outputPtr);
        outputPtr->clear();
        for (auto i = inputPtr->begin(); i != inputPtr->end(); ++i)
            mapSelf(*i, *outputPtr, minLength);
    }
    if (nextGenome != tgtGenome) {
        std::swap(inputPtr, outputPtr);
        outputPtr->clear();
        mapRecursiveDown(*inputPtr, *outputPtr, tgtGenome, onPath, doDupes, minLength, al);
    }
    if (outputPtr != &results)
        results = *outputPtr;
    results.sort(LessSourcePtr());
    results.unique(EqualToPtr());
    return results.size();
}

// halSegmentMapper.cpp:263-330
static size_t mapSelf(MSegPtr mappedSeg, MSegList &results, i64 minLength) {
    size_t added = 0;
    if (mappedSeg->isTop()) {
        const SegIt &top = mappedSeg->tgt;
        SegIt topCopy = top;
        do {
            MSegPtr newMappedSeg = newMSeg(mappedSeg->src, topCopy);
            results.push_back(newMappedSeg);
            ++added;
            if (topCopy.hasNextParalogy())
                topCopy.toNextParalogy();
        } while (topCopy.hasNextParalogy() && topCopy.getLength() >= minLength && topCopy.idx != top.idx);
    } else if (mappedSeg->tgt.G().parent >= 0) {
        i64 rightCutoff = mappedSeg->getEndPosition();
        const SegIt bottom = mappedSeg->tgt;
        i64 startOffset = bottom.so, endOffset = bottom.eo;
        SegIt top;
        top.toParseUp(bottom);
        do {
            SegIt topNew = top;
            SegIt bottomBack = bottom;
            bottomBack.toParseDown(topNew);
            i64 startBack = bottomBack.so, endBack = bottomBack.eo;
            SegIt newSource = mappedSeg->src;
            i64 startDelta = startBack - startOffset, endDelta = endBack - endOffset;
            newSource.slice(newSource.so + startDelta, newSource.eo + endDelta);
            MSegPtr newMappedSeg = newMSeg(newSource, topNew);
            added += mapSelf(newMappedSeg, results, minLength);
            if (top.getEndPosition() != rightCutoff)
                top.toRight(rightCutoff);
            else
                break;
        } while (true);
    }
    return added;
}

// halSegmentMapper.cpp:20, :332-357
enum OverlapCat { Same, Disjoint, AContainsB, BContainsA, AOverlapsLeftOfB, BOverlapsLeftOfA };
static OverlapCat slowOverlap(const SegIt &sA, const SegIt &sB) {
    i64 startA = sA.getStartPosition(), endA = sA.getEndPosition();
    i64 startB = sB.getStartPosition(), endB = sB.getEndPosition();
    if (startA > endA)
        std::swap(startA, endA);
    if (startB > endB)
        std::swap(startB, endB);
    if (endA < startB || startA > endB)
        return Disjoint;
    else if (startA == startB && endA == endB)
        return Same;
    else if (startA >= startB && endA <= endB)
        return BContainsA;
    else if (startB >= startA && endB <= endA)
        return AContainsB;
    else if (startA <= startB && endA < endB)
        return AOverlapsLeftOfB;
    return BOverlapsLeftOfA;
}

// halSegmentMapper.cpp:359-395
static void getOverlapBounds(MSegPtr &seg, MSegSet &results, MSegSet::iterator &leftBound, MSegSet::iterator &rightBound) {
    if (results.size() <= 2) {
        leftBound = results.begin();
        rightBound = results.end();
    } else {
        MSegSet::iterator i = results.lower_bound(seg);
        leftBound = i;
        if (leftBound != results.begin())
            --leftBound;
        MSegSet::iterator iprev;
        MSegSet::key_compare resLess = results.key_comp();
        while (leftBound != results.begin()) {
            iprev = leftBound;
            --iprev;
            if (leftBound == results.end() || !resLess(*iprev, *leftBound))
                leftBound = iprev;
            else
                break;
        }
        for (; leftBound != results.begin(); --leftBound) {
            if (leftBound != results.end() && slowOverlap(seg->tgt, (*leftBound)->tgt) == Disjoint)
                break;
        }
        rightBound = i;
        if (rightBound != results.end()) {
            for (++rightBound; rightBound != results.end(); ++rightBound) {
                if (slowOverlap(seg->tgt, (*rightBound)->tgt) == Disjoint)
                    break;
            }
        }
    }
}

// halSegmentMapper.cpp:397-473
static void clipAagainstB(MSegPtr segA, MSegPtr segB, OverlapCat, std::vector<MSegPtr> &clippedSegs) {
    i64 startA = segA->getStartPosition(), endA = segA->getEndPosition();
    i64 startB = segB->getStartPosition(), endB = segB->getEndPosition();
    if (startA > endA)
        std::swap(startA, endA);
    if (startB > endB)
        std::swap(startB, endB);
    MSegPtr left = segA;
    MSegPtr middle = MSegPtr(new MSeg(*segA));
    MSegPtr right;
    i64 startO = segA->getStartOffset(), endO = segA->getEndOffset();
    i64 length = segA->getLength();
    i64 leftSize = std::max((i64)0, startB - startA);
    i64 rightSize = std::max((i64)0, endA - endB);
    i64 middleSize = length - leftSize - rightSize;
    if (rightSize > 0)
        right = MSegPtr(new MSeg(*segA));
    i64 leftSlice = 0, rightSlice = 0;
    if (leftSize > 0) {
        leftSlice = 0;
        rightSlice = length - leftSize;
        if (left->getReversed())
            std::swap(leftSlice, rightSlice);
        left->slice(startO + leftSlice, endO + rightSlice);
    } else {
        middle = segA;
    }
    leftSlice = leftSize;
    rightSlice = rightSize;
    if (middle->getReversed())
        std::swap(leftSlice, rightSlice);
    middle->slice(startO + leftSlice, endO + rightSlice);
    if (middle.get() != segA.get())
        clippedSegs.push_back(middle);
    if (rightSize > 0) {
        leftSlice = leftSize + middleSize;
        rightSlice = 0;
        if (right->getReversed())
            std::swap(leftSlice, rightSlice);
        right->slice(startO + leftSlice, endO + rightSlice);
        clippedSegs.push_back(right);
    }
}

// halSegmentMapper.cpp:475-520
static void insertAndBreakOverlaps(MSegPtr seg, MSegSet &results) {
    MSegList inputSegs;
    std::vector<MSegPtr> clippedSegs;
    MSegSet::iterator leftBound, rightBound;
    getOverlapBounds(seg, results, leftBound, rightBound);
    bool leftBegin = leftBound == results.begin();
    MSegList::iterator inputIt;
    OverlapCat oc;
    inputSegs.push_back(seg);
    MSegSet::iterator resIt;
    for (resIt = leftBound; resIt != rightBound; ++resIt) {
        for (inputIt = inputSegs.begin(); inputIt != inputSegs.end(); ++inputIt) {
            oc = slowOverlap((*inputIt)->tgt, (*resIt)->tgt);
            if (oc == AContainsB || oc == AOverlapsLeftOfB || oc == BOverlapsLeftOfA) {
                clippedSegs.clear();
                clipAagainstB(*inputIt, *resIt, oc, clippedSegs);
                inputSegs.insert(inputSegs.end(), clippedSegs.begin(), clippedSegs.end());
            }
        }
    }
    for (inputIt = inputSegs.begin(); inputIt != inputSegs.end(); ++inputIt) {
        resIt = leftBegin ? results.begin() : leftBound;
        for (; resIt != rightBound; ++resIt) {
            oc = slowOverlap((*resIt)->tgt, (*inputIt)->tgt);
            if (oc == AContainsB) {
                clippedSegs.clear();
                clipAagainstB(*resIt, *inputIt, oc, clippedSegs);
                results.insert(clippedSegs.begin(), clippedSegs.end());
            }
        }
    }
    results.insert(inputSegs.begin(), inputSegs.end());
}

// halSegmentMapper.cpp:525-576
static size_t mapRecursiveParalogies(int srcGenome, MSegList &input, MSegList &results, const std::set<int> &onPath,
                                     int coalescenceLimit, i64 minLength, const Alignment &al) {
    if (input.empty()) {
        results = input;
        return 0;
    }
    int curGenome = (*input.begin())->getGenome();
    if (curGenome == coalescenceLimit) {
        results = input;
        return 0;
    }
    int nextGenome = al.genomes[(size_t)curGenome].parent;
    if (nextGenome < 0)
        throw std::runtime_error("Hit root genome when attempting to map paralogies");
    MSegList paralogs;
    for (auto i = input.begin(); i != input.end(); ++i)
        mapSelf(*i, paralogs, minLength);
    if (nextGenome != coalescenceLimit) {
        MSegList nextSegments;
        for (auto i = input.begin(); i != input.end(); ++i)
            mapUp(*i, nextSegments, true, minLength);
        mapRecursiveParalogies(srcGenome, nextSegments, results, onPath, coalescenceLimit, minLength, al);
    }
    MSegList paralogsMappedToSrc;
    mapRecursiveDown(paralogs, paralogsMappedToSrc, srcGenome, onPath, false, minLength, al);
    results.splice(results.begin(), paralogsMappedToSrc);
    results.sort(LessSourcePtr());
    results.unique(EqualToPtr());
    return results.size();
}

// halSegmentMapper.cpp:578-637
static size_t mapSource(const SegIt &source, MSegSet &results, int tgtGenome, const std::set<int> *genomesOnPath, bool doDupes,
                        i64 minLength, int coalescenceLimit, int mrca) {
    const Alignment &al = *source.al;
    MSegPtr newMappedSeg = newMSeg(source, source);
    MSegList input;
    input.push_back(newMappedSeg);
    MSegList output;
    const std::set<int> &onPath = *genomesOnPath;
    MSegList upResults;
    if (source.g != mrca)
        mapRecursiveUp(input, upResults, mrca, minLength, al);
    else
        upResults = input;
    MSegList paralogResults;
    if (mrca != coalescenceLimit && doDupes)
        mapRecursiveParalogies(mrca, upResults, paralogResults, onPath, coalescenceLimit, minLength, al);
    else
        paralogResults = upResults;
    if (tgtGenome != mrca)
        mapRecursiveDown(paralogResults, output, tgtGenome, onPath, doDupes, minLength, al);
    else
        output = paralogResults;
    for (auto outIt = output.begin(); outIt != output.end(); ++outIt)
        insertAndBreakOverlaps(*outIt, results);
    return output.size();
}

// halSegmentMapper.cpp:639-670
size_t halMapSegment(const SegIt &source, MSegSet &out, int tgtGenome, const std::set<int> *genomesOnPath, bool doDupes,
                     i64 minLength, int coalescenceLimit, int mrca) {
    const Alignment &al = *source.al;
    if (mrca < 0) {
        std::set<int> in;
        in.insert(source.g);
        in.insert(tgtGenome);
        mrca = getLowestCommonAncestor(al, in);
    }
    if (coalescenceLimit < 0)
        coalescenceLimit = mrca;
    std::set<int> pathSet;
    if (genomesOnPath == nullptr) {
        std::set<int> in;
        in.insert(tgtGenome);
        in.insert(mrca);
        getGenomesInSpanningTree(al, in, pathSet);
        genomesOnPath = &pathSet;
    }
    return mapSource(source, out, tgtGenome, genomesOnPath, doDupes, minLength, coalescenceLimit, mrca);
}

} // namespace orc
