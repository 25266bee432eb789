// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp and oracle_blockviz.hpp).
#include "oracle_blockviz.hpp"
#include <deque>
#include <limits>

namespace orc {

i64 BlockMapper::maxAdjScan = 1; // halBlockMapper.cpp:20

// ---- MappedSegment members the adjacency code uses (api/impl/halMappedSegment.cpp:63-73) ----
static void flip(MSeg &m) {
    std::swap(m.src, m.tgt);
}
static void fullReverse(MSeg &m) {
    m.src.slice(m.src.eo, m.src.so);
    m.src.toReverse();
    m.tgt.slice(m.tgt.eo, m.tgt.so);
    m.tgt.toReverse();
}

// halBlockMapper.cpp:29-34
void BlockMapper::erase() {
    segSet.clear();
    adjSet.clear();
    downwardPath.clear();
    upwardPath.clear();
}

// halBlockMapper.cpp:36-77
void BlockMapper::init(const Alignment *alignment, int refGenome_, int queryGenome_, i64 absRefFirst_, i64 absRefLast_, bool targetReversed_,
                       bool doDupes_, i64 minLength_, bool mapTargetAdjacencies, int coalescenceLimit_) {
    erase();
    al = alignment;
    absRefFirst = absRefFirst_;
    absRefLast = absRefLast_;
    targetReversed = targetReversed_;
    doDupes = doDupes_;
    minLength = minLength_;
    mapAdj = mapTargetAdjacencies;
    refGenome = refGenome_;
    refSequence = al->genomes[(size_t)refGenome].seqBySite(absRefFirst);
    queryGenome = queryGenome_;
    std::set<int> inputSet;
    inputSet.insert(refGenome);
    inputSet.insert(queryGenome);
    mrca = getLowestCommonAncestor(*al, inputSet);
    coalescenceLimit = coalescenceLimit_ < 0 ? mrca : coalescenceLimit_;
    inputSet.clear();
    inputSet.insert(queryGenome);
    inputSet.insert(coalescenceLimit);
    getGenomesInSpanningTree(*al, inputSet, downwardPath);
    inputSet.clear();
    inputSet.insert(refGenome);
    inputSet.insert(coalescenceLimit);
    getGenomesInSpanningTree(*al, inputSet, upwardPath);
}

// halBlockMapper.cpp:79-119
void BlockMapper::map() {
    const Genome &R = al->genomes[(size_t)refGenome];
    SegIt refSeg;
    refSeg.al = al;
    refSeg.g = refGenome;
    i64 lastIndex;
    if (mrca == refGenome && refGenome != queryGenome) {
        refSeg.top = false;
        lastIndex = R.numBot;
    } else {
        refSeg.top = true;
        lastIndex = R.numTop;
    }
    if (lastIndex > 0) { // (an iterator over an empty array has nothing to stand on)
        refSeg.toSite(absRefFirst, false);
        i64 startOffset = absRefFirst - refSeg.getStartPosition();
        i64 endOffset = 0;
        if (absRefLast <= refSeg.getEndPosition())
            endOffset = refSeg.getEndPosition() - absRefLast;
        refSeg.slice(startOffset, endOffset);
        while (refSeg.idx < lastIndex && refSeg.getStartPosition() <= absRefLast) {
            if (targetReversed)
                refSeg.toReverseInPlace();
            halMapSegment(refSeg, segSet, queryGenome, &downwardPath, doDupes, minLength, coalescenceLimit, mrca);
            if (targetReversed)
                refSeg.toReverseInPlace();
            refSeg.toRight(absRefLast);
        }
    }
    if (mapAdj) {
        for (MSegSet::const_iterator i = segSet.begin(); i != segSet.end(); ++i)
            if (adjSet.find(*i) == adjSet.end())
                mapAdjacencies(i);
    }
}

// halBlockMapper.cpp:247-271
SegIt BlockMapper::makeIterator(const MSegPtr &mappedSegment, i64 &minIndex, i64 &maxIndex) {
    SegIt segIt;
    segIt.al = mappedSegment->tgt.al;
    segIt.g = mappedSegment->getGenome();
    segIt.top = mappedSegment->isTop();
    segIt.idx = mappedSegment->tgt.idx;
    const Sequence *seq = segIt.getSequence();
    if (segIt.top) {
        minIndex = seq->topStart;
        maxIndex = minIndex + seq->numTop;
    } else {
        minIndex = seq->botStart;
        maxIndex = minIndex + seq->numBot;
    }
    if (mappedSegment->getReversed())
        segIt.toReverse();
    segIt.slice(mappedSegment->getStartOffset(), mappedSegment->getEndOffset());
    return segIt;
}

// halBlockMapper.cpp:273-329
bool BlockMapper::cutByNext(SegIt &query, const SegIt &nextSeg, bool right) {
    bool wasCut = false;
    if (query.idx == nextSeg.idx) {
        i64 so1 = query.so;
        i64 eo1 = query.eo;
        if (query.rev)
            std::swap(so1, eo1);
        i64 so2 = nextSeg.rev ? nextSeg.eo : nextSeg.so;
        if (right) {
            if (so1 >= so2) {
                wasCut = true;
            } else {
                i64 e1 = std::max(query.getEndPosition(), query.getStartPosition());
                i64 s2 = std::min(nextSeg.getEndPosition(), nextSeg.getStartPosition());
                if (e1 >= s2) {
                    i64 delta = 1 + e1 - s2;
                    i64 newEndOffset = eo1 + delta;
                    i64 newStartOffset = so1;
                    if (query.rev)
                        std::swap(newEndOffset, newStartOffset);
                    query.slice(newStartOffset, newEndOffset);
                }
            }
        } else {
            i64 s1 = std::min(query.getEndPosition(), query.getStartPosition());
            i64 e1 = std::max(query.getEndPosition(), query.getStartPosition());
            i64 e2 = std::max(nextSeg.getEndPosition(), nextSeg.getStartPosition());
            if (e1 <= e2) {
                wasCut = true;
            } else {
                if (s1 <= e2) {
                    i64 delta = 1 + e2 - s1;
                    i64 newStartOffset = so1 + delta;
                    i64 newEndOffset = eo1;
                    if (query.rev)
                        std::swap(newEndOffset, newStartOffset);
                    query.slice(newStartOffset, newEndOffset);
                }
            }
        }
    }
    return wasCut;
}

// halBlockMapper.cpp:121-245
void BlockMapper::mapAdjacencies(MSegSet::const_iterator segIt) {
    MSegPtr mappedQuerySeg(*segIt);
    i64 maxIndex, minIndex;
    SegIt queryIt = makeIterator(mappedQuerySeg, minIndex, maxIndex);
    MSegSet backResults;
    MSegSet::const_iterator segNext = segIt;
    if (queryIt.rev)
        segNext = segNext == segSet.begin() ? segSet.end() : --segNext;
    else
        ++segNext;

    i64 iter = 0;
    queryIt.toRight();
    while (queryIt.idx >= minIndex && queryIt.idx < maxIndex && iter < maxAdjScan) {
        bool wasCut = false;
        if (segNext != segSet.end())
            wasCut = cutByNext(queryIt, (*segNext)->tgt, !queryIt.rev);
        if (wasCut)
            break;
        size_t backSize = backResults.size();
        halMapSegment(queryIt, backResults, refGenome, &upwardPath, doDupes, minLength, -1, -1);
        if (backResults.size() > backSize)
            break;
        queryIt.toRight();
        ++iter;
    }

    queryIt = makeIterator(mappedQuerySeg, minIndex, maxIndex);
    MSegSet::const_iterator segPrev = segIt;
    if (queryIt.rev)
        ++segPrev;
    else
        segPrev = segPrev == segSet.begin() ? segSet.end() : --segPrev;
    iter = 0;
    queryIt.toLeft();
    while (queryIt.idx >= minIndex && queryIt.idx < maxIndex && iter < maxAdjScan) {
        bool wasCut = false;
        if (segPrev != segSet.end())
            wasCut = cutByNext(queryIt, (*segPrev)->tgt, queryIt.rev);
        if (wasCut)
            break;
        size_t backSize = backResults.size();
        halMapSegment(queryIt, backResults, refGenome, &upwardPath, doDupes, minLength, -1, -1);
        if (backResults.size() > backSize)
            break;
        queryIt.toLeft();
        ++iter;
    }

    MSegSet outSet;
    // flip the results (in place, as the reference does with the members of its set) and copy back to the main set
    for (MSegSet::iterator i = backResults.begin(); i != backResults.end(); ++i) {
        MSegPtr mseg(*i);
        if (mseg->tgt.getSequence() == refSequence) {
            flip(*mseg);
            if (mseg->src.rev)
                fullReverse(*mseg);
            MSegSet::const_iterator j = segSet.lower_bound(*i);
            bool overlaps = false;
            if (j != segSet.begin())
                --j;
            for (size_t count = 0; count < 3 && j != segSet.end() && !overlaps; ++count, ++j) {
                overlaps = mseg->tgt.overlaps((*j)->getStartPosition()) || mseg->tgt.overlaps((*j)->getEndPosition()) ||
                           (*j)->tgt.overlaps(mseg->getStartPosition()) || (*j)->tgt.overlaps(mseg->getEndPosition());
            }
            if (!overlaps)
                outSet.insert(mseg);
        }
    }

    // clean up dupes before adding to output
    for (MSegSet::iterator i = outSet.begin(); i != outSet.end();) {
        MSegSet::iterator j = i;
        ++j;
        while (j != outSet.end() &&
               ((*j)->getStartPosition() == (*i)->getStartPosition() || (*j)->getEndPosition() == (*i)->getStartPosition()))
            ++j;
        MSegSet::iterator best = i;
        i64 best_delta = std::numeric_limits<i64>::max();
        for (MSegSet::iterator k = i; k != j; ++k) {
            i64 delta = std::min(std::abs((*k)->src.getStartPosition() - (*segIt)->src.getStartPosition()),
                                 std::abs((*k)->src.getEndPosition() - (*segIt)->src.getStartPosition()));
            if (delta < best_delta) {
                best_delta = delta;
                best = k;
            }
        }
        segSet.insert(*best);
        adjSet.insert(*best);
        i = j;
    }
}

// ---------------------------------------------------------------------------------------------
// blockViz/impl/halBlockViz.cpp

static void reverseComplementString(std::string &s) { // api/impl/halCommon.cpp:55-83 (no gaps occur here)
    auto comp = [](char c) {
        switch (c) {
        case 'A': return 'T';
        case 'a': return 't';
        case 'C': return 'G';
        case 'c': return 'g';
        case 'G': return 'C';
        case 'g': return 'c';
        case 'T': return 'A';
        case 't': return 'a';
        default: return c;
        }
    };
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i)
        r[s.size() - 1 - i] = comp(s[i]);
    s.swap(r);
}

static void getSubString(const Genome &G, const Sequence &S, std::string &out, i64 start, i64 length) { // Sequence::getSubString
    static const char unpack[16] = {'a', 'c', 'g', 't', 'n', '\0', '\0', '\0', 'A', 'C', 'G', 'T', 'N', '\0', '\0', '\0'};
    out.resize((size_t)length);
    for (i64 k = 0; k < length; ++k) {
        const i64 pos = S.start + start + k;
        if (G.dna.empty()) {
            out[(size_t)k] = 'N';
            continue;
        }
        const u8 b = G.dna[(size_t)(pos >> 1)];
        out[(size_t)k] = unpack[(pos & 1) ? (b & 0x0F) : (b >> 4)];
    }
}

// halBlockViz.cpp:832-905
static void readBlock(const Alignment &al, VizBlock &cur, std::vector<MSegPtr> &fragments, bool getSequenceString, const std::string &genomeName) {
    MSegPtr firstQuerySeg = fragments.front();
    MSegPtr lastQuerySeg = fragments.back();
    const SegIt &firstRefSeg = firstQuerySeg->src;
    const SegIt &lastRefSeg = lastQuerySeg->src;
    const Sequence *qSequence = firstQuerySeg->tgt.getSequence();
    const Sequence *tSequence = firstRefSeg.getSequence();
    std::string seqBuffer = qSequence->name;
    size_t prefix = seqBuffer.find(genomeName + '.') != 0 ? 0 : genomeName.length() + 1;
    cur.qChrom = seqBuffer.substr(prefix);
    cur.tStart = std::min(std::min(firstRefSeg.getStartPosition(), firstRefSeg.getEndPosition()),
                          std::min(lastRefSeg.getStartPosition(), lastRefSeg.getEndPosition()));
    cur.tStart -= tSequence->start;
    cur.qStart = std::min(std::min(firstQuerySeg->getStartPosition(), firstQuerySeg->getEndPosition()),
                          std::min(lastQuerySeg->getStartPosition(), lastQuerySeg->getEndPosition()));
    cur.qStart -= qSequence->start;
    i64 tEnd = std::max(std::max(firstRefSeg.getStartPosition(), firstRefSeg.getEndPosition()),
                        std::max(lastRefSeg.getStartPosition(), lastRefSeg.getEndPosition()));
    tEnd -= tSequence->start;
    cur.size = 1 + tEnd - cur.tStart;
    cur.strand = firstQuerySeg->getReversed() ? '-' : '+';
    if (getSequenceString) {
        getSubString(al.genomes[(size_t)firstQuerySeg->getGenome()], *qSequence, cur.qSequence, cur.qStart, cur.size);
        getSubString(al.genomes[(size_t)firstRefSeg.g], *tSequence, cur.tSequence, cur.tStart, cur.size);
        if (cur.strand == '-')
            reverseComplementString(cur.qSequence);
    }
}

// halBlockViz.cpp:1072-1178
static void chainReferenceParalogies(MSegSet &segMap, i64 absStart, i64 absEnd, MSegSet &outParalogies, double min_chain_pct = 0.025) {
    (void)absStart;
    (void)absEnd;
    std::vector<std::vector<MSegSet::iterator>> chains;
    std::vector<i64> chain_sizes;
    std::deque<i64> chain_stack;
    std::vector<MSegSet::iterator> filtered_paralogies;

    for (MSegSet::iterator i = segMap.begin(); i != segMap.end();) {
        MSegSet::iterator j = i;
        ++j;
        i64 copies = 1;
        while (j != segMap.end() &&
               ((*j)->getStartPosition() == (*i)->getStartPosition() || (*j)->getEndPosition() == (*i)->getStartPosition())) {
            ++j;
            ++copies;
        }
        i64 best_score = -(i64)std::numeric_limits<int32_t>::max();
        i64 best_stack_idx = -1;
        MSegSet::iterator best = segMap.end();
        MSegSet::iterator leftmost = segMap.end();
        i64 left_src_pos = std::numeric_limits<i64>::max();
        for (MSegSet::iterator k = i; k != j; ++k) {
            for (i64 csi = (i64)chain_stack.size() - 1; csi >= 0; --csi) {
                MSegSet::iterator &chain_back = chains[(size_t)chain_stack[(size_t)csi]].back();
                i64 src_delta = (*k)->src.getStartPosition() - (*chain_back)->src.getEndPosition();
                if ((*k)->getReversed())
                    src_delta = -src_delta;
                i64 tgt_delta = (*k)->getStartPosition() - (*chain_back)->getEndPosition();
                if (src_delta >= 0 && tgt_delta >= 0) {
                    i64 score = chain_sizes[(size_t)chain_stack[(size_t)csi]] * 2 - tgt_delta - src_delta;
                    if (score > best_score) {
                        best_stack_idx = csi;
                        best_score = score;
                        best = k;
                    }
                }
            }
            i64 mpos = std::min((*k)->getStartPosition(), (*k)->getEndPosition());
            if (mpos < left_src_pos) {
                left_src_pos = mpos;
                leftmost = k;
            }
        }
        if (best_stack_idx < 0) {
            best = leftmost;
            chains.push_back({best});
            chain_sizes.push_back((*best)->getLength());
            chain_stack.push_back((i64)chains.size() - 1);
        } else {
            chains[(size_t)chain_stack[(size_t)best_stack_idx]].push_back(best);
            chain_sizes[(size_t)chain_stack[(size_t)best_stack_idx]] += (*best)->getLength();
            while ((i64)chain_stack.size() - 1 > best_stack_idx)
                chain_stack.pop_back();
        }
        if (copies > 1) {
            for (MSegSet::iterator k = i; k != j; ++k) {
                outParalogies.insert(*k);
                if (k != best)
                    filtered_paralogies.push_back(k);
            }
        }
        i = j;
    }
    for (size_t i = 0; i < filtered_paralogies.size(); ++i)
        segMap.erase(filtered_paralogies[i]);
    i64 total_chain_size = 0;
    for (size_t chain = 0; chain < chains.size(); ++chain)
        total_chain_size += chain_sizes[chain];
    for (size_t chain = 0; chain < chains.size(); ++chain) {
        double chain_pct = (double)chain_sizes[chain] / (double)total_chain_size;
        if (chain_pct < min_chain_pct)
            for (size_t ci = 0; ci < chains[chain].size(); ++ci)
                segMap.erase(chains[chain][ci]);
    }
}

// halBlockViz.cpp:944-1052
static std::vector<VizTargetDupe> processTargetDupes(MSegSet &paraSet) {
    std::vector<std::pair<std::set<i64>, i64>> dupe_lists;
    for (MSegSet::iterator i = paraSet.begin(); i != paraSet.end();) {
        MSegSet::iterator j = i;
        ++j;
        while (j != paraSet.end() &&
               ((*j)->getStartPosition() == (*i)->getStartPosition() || (*j)->getEndPosition() == (*i)->getStartPosition()))
            ++j;
        std::set<i64> dupe_starts;
        for (MSegSet::iterator k = i; k != j; ++k)
            dupe_starts.insert((*k)->src.getStartPosition());
        dupe_lists.push_back(std::make_pair(dupe_starts, (i64)(*i)->getLength()));
        i = j;
    }
    std::sort(dupe_lists.begin(), dupe_lists.end(),
              [](const std::pair<std::set<i64>, i64> &d1, const std::pair<std::set<i64>, i64> &d2) { return *d1.first.begin() < *d2.first.begin(); });
    for (size_t i = 0; i < dupe_lists.size(); ++i) {
        if (dupe_lists[i].second <= 0)
            continue;
        for (size_t j = i + 1; j < dupe_lists.size(); ++j) {
            bool merged = false;
            if (dupe_lists[j].first.size() == dupe_lists[i].first.size()) {
                std::set<i64>::iterator k1 = dupe_lists[i].first.begin();
                std::set<i64>::iterator k2 = dupe_lists[j].first.begin();
                i64 min_extension = std::numeric_limits<i64>::max();
                for (; k1 != dupe_lists[i].first.end(); ++k1, ++k2) {
                    i64 left_overlap = -1;
                    if (*k2 >= *k1) {
                        left_overlap = (*k1 + dupe_lists[i].second) - *k2;
                        if (left_overlap > 0)
                            left_overlap = std::min(left_overlap, dupe_lists[j].second);
                    }
                    i64 right_extension = left_overlap < 0 ? -1 : left_overlap - dupe_lists[j].second;
                    min_extension = std::min(min_extension, right_extension);
                }
                if (min_extension == 0) {
                    dupe_lists[j].second = 0;
                } else if (min_extension > 0) {
                    dupe_lists[i].second += min_extension;
                    dupe_lists[j].second -= min_extension;
                }
                merged = min_extension >= 0;
            }
            if (!merged)
                break;
        }
    }
    std::vector<VizTargetDupe> out;
    i64 cur_id = 0;
    const Sequence *chrom = (*paraSet.begin())->src.getSequence();
    i64 prev = -1;
    for (size_t i = 0; i < dupe_lists.size(); ++i) {
        if (dupe_lists[i].second == 0)
            continue;
        VizTargetDupe dupe;
        if (prev >= 0) {
            i64 prev_end = *dupe_lists[(size_t)prev].first.begin() + dupe_lists[(size_t)prev].second;
            if (*dupe_lists[i].first.begin() > prev_end)
                ++cur_id;
        }
        dupe.id = cur_id;
        dupe.qChrom = chrom->name;
        for (std::set<i64>::iterator j = dupe_lists[i].first.begin(); j != dupe_lists[i].first.end(); ++j)
            dupe.ranges.push_back(std::make_pair(*j - chrom->start, dupe_lists[i].second));
        out.push_back(dupe);
        prev = (i64)i;
    }
    return out;
}

// halBlockViz.cpp:759-830
static VizResults readBlocks(const Alignment &al, int tGenome, const Sequence *tSequence, i64 absStart, i64 absEnd, bool tReversed, int qGenome,
                             bool getSequenceString, bool doDupes, bool doTargetDupes, bool doAdjes, int coalescenceLimit,
                             bool coalescenceLimitGiven) {
    (void)tSequence;
    const std::string qGenomeName = al.genomes[(size_t)qGenome].name;
    BlockMapper blockMapper;
    if (qGenome == tGenome && !coalescenceLimitGiven)
        blockMapper.init(&al, tGenome, qGenome, absStart, absEnd, tReversed, doDupes, 0, doAdjes, al.root());
    else
        blockMapper.init(&al, tGenome, qGenome, absStart, absEnd, tReversed, doDupes, 0, doAdjes, coalescenceLimitGiven ? coalescenceLimit : -1);
    blockMapper.map();
    MSegSet paraSet;
    MSegSet &segMap = blockMapper.segSet;
    if (doDupes && qGenome != tGenome)
        chainReferenceParalogies(segMap, absStart, absEnd, paraSet);
    std::vector<MSegPtr> fragments;
    std::set<i64> queryCutSet, targetCutSet;
    targetCutSet.insert(blockMapper.absRefFirst);
    targetCutSet.insert(blockMapper.absRefLast);
    VizResults results;
    for (MSegSet::iterator segMapIt = segMap.begin(); segMapIt != segMap.end(); ++segMapIt) {
        VizBlock cur;
        extractSegment(segMapIt, paraSet, fragments, &segMap, targetCutSet, queryCutSet);
        readBlock(al, cur, fragments, getSequenceString, qGenomeName);
        results.mappedBlocks.push_back(cur);
    }
    if (!paraSet.empty())
        results.targetDupeBlocks = processTargetDupes(paraSet);
    if (!doTargetDupes)
        results.targetDupeBlocks.clear();
    return results;
}

// halBlockViz.cpp:243-330
VizResults getBlocksInTargetRange(const Alignment &al, int qGenome, int tGenome, const std::string &tChrom, i64 tStart, i64 tEnd, bool tReversed,
                                  bool getSequenceString, int dupMode, bool mapBackAdjacencies, int coalescenceLimit) {
    const i64 rangeLength = tEnd - tStart;
    if (rangeLength < 0)
        throw std::runtime_error("halGetBlocksInTargetRange invalid query range [" + std::to_string(tStart) + "," + std::to_string(tEnd) + ")");
    if (tReversed && mapBackAdjacencies)
        throw std::runtime_error("halGetBlocksInTargetRange tReversed can only be set when mapBackAdjacencies is 0");
    if (tReversed && dupMode == VIZ_QUERY_AND_TARGET_DUPS)
        throw std::runtime_error("tReversed cannot be set in conjunction with dupMode=HAL_QUERY_AND_TARGET_DUPS");
    const Sequence *tSequence = al.genomes[(size_t)tGenome].seqByName(tChrom);
    if (tSequence == nullptr)
        throw std::runtime_error("Unable to locate sequence " + tChrom + " in genome " + al.genomes[(size_t)tGenome].name);
    const i64 myEnd = tEnd > 0 ? tEnd : tSequence->length;
    const i64 absStart = tSequence->start + tStart;
    const i64 absEnd = tSequence->start + myEnd - 1;
    if (absStart > absEnd)
        throw std::runtime_error("halGetBlocksInTargetRange invalid range");
    if (absEnd > tSequence->start + tSequence->length - 1)
        throw std::runtime_error("halGetBlocksInTargetRange target end position outside of target sequence");
    return readBlocks(al, tGenome, tSequence, absStart, absEnd, tReversed, qGenome, getSequenceString, dupMode != VIZ_NO_DUPS,
                      dupMode == VIZ_QUERY_AND_TARGET_DUPS, mapBackAdjacencies, coalescenceLimit, coalescenceLimit >= 0);
}

// blockViz/tests/blockVizTest.cpp:103-113
void printVizResults(std::ostream &os, const VizResults &r, bool withSequence) {
    for (const VizBlock &b : r.mappedBlocks) {
        os << "chr:" << b.qChrom << ", tSt:" << b.tStart << ", qSt:" << b.qStart << ", size:" << b.size << ", strand:" << b.strand << ": tgt : "
           << (withSequence ? b.tSequence.substr(0, 10) : std::string("(null)")) << " query: "
           << (withSequence ? b.qSequence.substr(0, 10) : std::string("(null)")) << "\n";
    }
    for (const VizTargetDupe &d : r.targetDupeBlocks) {
        os << "tDupe id:" << d.id << " qCrhom:" << d.qChrom << "\n";
        for (const auto &tr : d.ranges)
            os << " tSt:" << tr.first << " size:" << tr.second << "\n";
    }
}

} // namespace orc
