// The blockViz query behind the C ABI: halGetBlocksInTargetRange (blockViz/impl/halBlockViz.cpp:243-330 -> readBlocks :759-830)
// = BlockMapper::init + map WITH adjacencies (liftover/impl/halBlockMapper.cpp:36-245) + chainReferenceParalogies (:1072-1178)
// + BlockMapper::extractSegment with the paralogy set and the target cut points (halBlockMapper.cpp:331-394) + readBlock
// (:832-905) + processTargetDupes (:944-1052).
//
// Where the work goes.  The graph chases are the GPU's, in two batches per call, however many ranges the call carries:
//   1. every range of the reference genome mapped to the query genome (the refined set of BlockMapper::getMap: the walk kernels
//      and the finishing kernel in blocks mode, hgx_liftover.hip);
//   2. for every member of those sets, the stretch of the query genome next to it on either side — what the segment iterator
//      of mapAdjacencies steps onto with toRight() / toLeft(): the rest of the member's segment or the whole neighbouring
//      segment — mapped back to the reference genome, as the pieces halMapSegment's walk leaves before insertAndBreakOverlaps
//      (blocks mode 2).  The map of a sub-range is the map of the range clipped (composed tables), so the cut that
//      cutByNext applies — which depends on what earlier adjacencies have put into the set — is applied to the pieces on the host.
// The sequential rest (the set that grows while it is walked, the chaining, the merging of fragments into blocks) runs on the
// host over those pieces, in forward coordinates: a piece is (target range, source range, two orientations); the reference's
// set order (MappedSegment::lessThan: target then source, fastComp on index and offsets, api/impl/halMappedSegment.cpp:36-61,
// 167-206) is the lexicographic order of (target low, target high, source low, source high).
#include "hgx_liftover_engine.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <set>
#include <stdexcept>

namespace hgx {

namespace {

// a mapped segment (api/inc/halMappedSegment.h:196-197) in forward coordinates: the side it "is" (getStartPosition, getGenome:
// the target of the mapping) and its source
struct Piece {
    int64_t tLo, tHi, sLo, sHi;
    bool tRev, sRev;
    int64_t tStart() const { return tRev ? tHi : tLo; } // SlicedSegment::getStartPosition
    int64_t tEnd() const { return tRev ? tLo : tHi; }
    int64_t sStart() const { return sRev ? sHi : sLo; }
    int64_t sEnd() const { return sRev ? sLo : sHi; }
    int64_t length() const { return tHi - tLo + 1; }
    bool overlaps(int64_t pos) const { return pos >= tLo && pos <= tHi; } // halSegmentIterator.cpp:86-108
};
struct PieceLess {
    bool operator()(const Piece &a, const Piece &b) const {
        if (a.tLo != b.tLo)
            return a.tLo < b.tLo;
        if (a.tHi != b.tHi)
            return a.tHi < b.tHi;
        if (a.sLo != b.sLo)
            return a.sLo < b.sLo;
        return a.sHi < b.sHi;
    }
};
typedef std::set<Piece, PieceLess> PieceSet;

// index of the segment of a tiling (start[] with its sentinel) that holds pos
int64_t segmentOf(const std::vector<int64_t> &start, int64_t pos) {
    return (int64_t)(std::upper_bound(start.begin(), start.end() - 1, pos) - start.begin()) - 1;
}

// insertAndBreakOverlaps over a whole set at once (api/impl/halSegmentMapper.cpp:475-520): every piece cut at every boundary
// of every piece whose target range overlaps its own, source side sliced in step (MappedSegment::slice); equal pieces once
void refineInto(std::vector<Piece> &pieces, PieceSet &out) {
    std::vector<int64_t> cuts; // piece starts and piece ends + 1
    cuts.reserve(2 * pieces.size());
    for (const Piece &p : pieces) {
        cuts.push_back(p.tLo);
        cuts.push_back(p.tHi + 1);
    }
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    // a cut inside p only counts when it is the boundary of a piece that overlaps p — every boundary strictly inside p's range
    // belongs to such a piece (its owner starts or ends inside p)
    for (const Piece &p : pieces) {
        int64_t lo = p.tLo;
        auto it = std::upper_bound(cuts.begin(), cuts.end(), p.tLo);
        for (;;) {
            const int64_t hi = (it != cuts.end() && *it <= p.tHi) ? *it - 1 : p.tHi;
            Piece q = p;
            q.tLo = lo;
            q.tHi = hi;
            if (p.tRev == p.sRev) {
                q.sLo = p.sLo + (lo - p.tLo);
                q.sHi = p.sLo + (hi - p.tLo);
            } else {
                q.sLo = p.sHi - (hi - p.tLo);
                q.sHi = p.sHi - (lo - p.tLo);
            }
            out.insert(q);
            if (hi == p.tHi)
                break;
            lo = hi + 1;
            ++it;
        }
    }
}

struct RangeJob { // one BlockMapper
    int64_t absFirst, absLast;
    PieceSet segSet, adjSet;
};

struct Mapper { // what BlockMapper::init fixes for all ranges of a call
    hgx_alignment *h;
    int ref, query, mrca, limit;
    bool doDupes, queryTop; // queryTop: the mapped segments' own side lies on the query genome's top tiling
    const GenomeTables *R, *Q;
};

// the member's records of a blocks-mode run -> pieces; target side = the run's target genome (record fields: include/hgx.h,
// hgx_liftover_opts.emit_blocks)
Piece pieceOf(const hgx_record &r, const GenomeTables &T) {
    Piece p;
    p.tLo = r.tgt_start + T.seqs[(size_t)r.tgt_seq].start;
    p.tHi = r.tgt_end - 1 + T.seqs[(size_t)r.tgt_seq].start;
    p.sLo = r.src_start;
    p.sHi = r.src_start + (r.tgt_end - r.tgt_start) - 1;
    p.tRev = r.tgt_reversed != 0;
    p.sRev = r.strand == '-';
    return p;
}

// the stretch of the query genome a segment iterator made from member m (BlockMapper::makeIterator, halBlockMapper.cpp:247-271)
// stands on after toRight() (right = true) or toLeft() (halSegmentIterator.cpp:177-238): the rest of m's segment on that side,
// or the whole neighbouring segment.  false: the neighbour lies outside m's sequence (:148, :178).  segIdx: its segment.
bool neighbour(const Mapper &M, const Piece &m, bool right, int64_t &lo, int64_t &hi, int64_t &segIdx) {
    const std::vector<int64_t> &start = M.queryTop ? M.Q->tStart : M.Q->bStart;
    const int64_t idx = segmentOf(start, m.tLo);
    const int64_t segLo = start[(size_t)idx], segHi = start[(size_t)idx + 1] - 1;
    // in forward coordinates: an unreversed iterator's toRight moves up, a reversed one's moves down; toLeft the other way
    const bool up = right != m.tRev;
    int64_t j = idx;
    if (up) {
        if (m.tHi < segHi) {
            lo = m.tHi + 1;
            hi = segHi;
        } else {
            j = idx + 1;
        }
    } else {
        if (m.tLo > segLo) {
            lo = segLo;
            hi = m.tLo - 1;
        } else {
            j = idx - 1;
        }
    }
    const int s = M.Q->seqIndexBySite(segLo);
    const SeqInfo &S = M.Q->seqs[(size_t)s];
    const int64_t minIndex = M.queryTop ? S.topStart : S.botStart, maxIndex = minIndex + (M.queryTop ? S.numTop : S.numBot);
    if (j < minIndex || j >= maxIndex)
        return false;
    if (j != idx) {
        lo = start[(size_t)j];
        hi = start[(size_t)j + 1] - 1;
    }
    segIdx = j;
    return true;
}

// BlockMapper::cutByNext (halBlockMapper.cpp:273-329) on the stretch [lo, hi] of segment segIdx against the set member `next`:
// true = nothing is left (the stretch begins inside or behind `next`: already mapped)
bool cutByNext(const Mapper &M, int64_t segIdx, bool queryRev, int64_t &lo, int64_t &hi, const Piece &next, bool right) {
    const std::vector<int64_t> &start = M.queryTop ? M.Q->tStart : M.Q->bStart;
    if (segmentOf(start, next.tLo) != segIdx)
        return false;
    (void)queryRev; // (so1 / eo1 of the reference are the forward offsets whatever the orientation: :280-285)
    if (right) {
        if (lo >= next.tLo) // so1 >= so2
            return true;
        if (hi >= next.tLo)
            hi = next.tLo - 1;
    } else {
        if (hi <= next.tHi)
            return true;
        if (lo <= next.tHi)
            lo = next.tHi + 1;
    }
    return false;
}

// BlockMapper::mapAdjacencies (halBlockMapper.cpp:121-245) for member `segIt` of J.segSet; back[2]: the pieces of the query
// genome's stretches to its right and left mapped back to the reference genome (unrefined, target side = reference genome)
void mapAdjacencies(const Mapper &M, RangeJob &J, PieceSet::const_iterator segIt, const std::vector<Piece> back[2], const int refSequence) {
    const Piece m = *segIt;
    std::vector<Piece> found;
    for (int side = 0; side < 2; ++side) {
        const bool right = side == 0;
        int64_t lo = 0, hi = -1, segIdx = -1;
        if (!neighbour(M, m, right, lo, hi, segIdx))
            continue;
        // the member the stretch is cut against: the next one in set order for the side the iterator moves up on (:129-134, :161-166)
        PieceSet::const_iterator other = segIt;
        const bool up = right != m.tRev;
        if (up) {
            ++other;
        } else {
            other = other == J.segSet.begin() ? J.segSet.end() : --other;
        }
        if (other != J.segSet.end() && cutByNext(M, segIdx, m.tRev, lo, hi, *other, up))
            continue;
        for (const Piece &b : back[side]) { // clip to what the cut left: source side = query genome
            const int64_t c = std::max(b.sLo, lo), d = std::min(b.sHi, hi);
            if (c > d)
                continue;
            Piece q = b;
            q.sLo = c;
            q.sHi = d;
            if (b.tRev == b.sRev) {
                q.tLo = b.tLo + (c - b.sLo);
                q.tHi = b.tLo + (d - b.sLo);
            } else {
                q.tLo = b.tHi - (d - b.sLo);
                q.tHi = b.tHi - (c - b.sLo);
            }
            found.push_back(q);
        }
    }
    PieceSet backResults; // the set both halMapSegment calls fill (:126): refined together
    refineInto(found, backResults);
    // flip the results and copy back to the main set (:187-212)
    PieceSet outSet;
    for (const Piece &b : backResults) {
        if (M.R->seqIndexBySite(b.tLo) != refSequence)
            continue;
        Piece f; // flip(): the reference side becomes the source
        f.tLo = b.sLo;
        f.tHi = b.sHi;
        f.sLo = b.tLo;
        f.sHi = b.tHi;
        f.tRev = b.sRev;
        f.sRev = b.tRev;
        if (f.sRev) { // fullReverse()
            f.sRev = false;
            f.tRev = !f.tRev;
        }
        PieceSet::const_iterator j = J.segSet.lower_bound(f);
        bool overlaps = false;
        if (j != J.segSet.begin())
            --j;
        for (size_t count = 0; count < 3 && j != J.segSet.end() && !overlaps; ++count, ++j)
            overlaps = f.overlaps(j->tStart()) || f.overlaps(j->tEnd()) || j->overlaps(f.tStart()) || j->overlaps(f.tEnd());
        if (!overlaps)
            outSet.insert(f);
    }
    // clean up dupes before adding to output (:214-243)
    for (PieceSet::iterator i = outSet.begin(); i != outSet.end();) {
        PieceSet::iterator j = i;
        ++j;
        while (j != outSet.end() && (j->tStart() == i->tStart() || j->tEnd() == i->tStart()))
            ++j;
        PieceSet::iterator best = i;
        int64_t bestDelta = std::numeric_limits<int64_t>::max();
        for (PieceSet::iterator k = i; k != j; ++k) {
            const int64_t delta = std::min(std::llabs(k->sStart() - m.sStart()), std::llabs(k->sEnd() - m.sStart()));
            if (delta < bestDelta) {
                bestDelta = delta;
                best = k;
            }
        }
        J.segSet.insert(*best);
        J.adjSet.insert(*best);
        i = j;
    }
}

// chainReferenceParalogies (halBlockViz.cpp:1072-1178): the query genome single copy — a greedy chaining of the members whose
// (query-side) ranges coincide; the others go to outParalogies and leave the set, and so do chains below min_chain_pct
void chainReferenceParalogies(PieceSet &segMap, PieceSet &outParalogies, double minChainPct = 0.025) {
    std::vector<std::vector<PieceSet::iterator>> chains;
    std::vector<int64_t> chainSizes;
    std::deque<int64_t> chainStack;
    std::vector<PieceSet::iterator> filtered;
    for (PieceSet::iterator i = segMap.begin(); i != segMap.end();) {
        PieceSet::iterator j = i;
        ++j;
        int64_t copies = 1;
        while (j != segMap.end() && (j->tStart() == i->tStart() || j->tEnd() == i->tStart())) {
            ++j;
            ++copies;
        }
        int64_t bestScore = -(int64_t)std::numeric_limits<int32_t>::max(), bestStackIdx = -1;
        PieceSet::iterator best = segMap.end(), leftmost = segMap.end();
        int64_t leftPos = std::numeric_limits<int64_t>::max();
        for (PieceSet::iterator k = i; k != j; ++k) {
            for (int64_t csi = (int64_t)chainStack.size() - 1; csi >= 0; --csi) {
                const Piece &chainBack = *chains[(size_t)chainStack[(size_t)csi]].back();
                int64_t srcDelta = k->sStart() - chainBack.sEnd();
                if (k->tRev)
                    srcDelta = -srcDelta;
                const int64_t tgtDelta = k->tStart() - chainBack.tEnd();
                if (srcDelta >= 0 && tgtDelta >= 0) {
                    const int64_t score = chainSizes[(size_t)chainStack[(size_t)csi]] * 2 - tgtDelta - srcDelta;
                    if (score > bestScore) {
                        bestStackIdx = csi;
                        bestScore = score;
                        best = k;
                    }
                }
            }
            if (k->tLo < leftPos) {
                leftPos = k->tLo;
                leftmost = k;
            }
        }
        if (bestStackIdx < 0) {
            best = leftmost;
            chains.push_back({best});
            chainSizes.push_back(best->length());
            chainStack.push_back((int64_t)chains.size() - 1);
        } else {
            chains[(size_t)chainStack[(size_t)bestStackIdx]].push_back(best);
            chainSizes[(size_t)chainStack[(size_t)bestStackIdx]] += best->length();
            while ((int64_t)chainStack.size() - 1 > bestStackIdx)
                chainStack.pop_back();
        }
        if (copies > 1)
            for (PieceSet::iterator k = i; k != j; ++k) {
                outParalogies.insert(*k);
                if (k != best)
                    filtered.push_back(k);
            }
        i = j;
    }
    for (PieceSet::iterator k : filtered)
        segMap.erase(k);
    int64_t total = 0;
    for (int64_t s : chainSizes)
        total += s;
    for (size_t c = 0; c < chains.size(); ++c)
        if ((double)chainSizes[c] / (double)total < minChainPct)
            for (PieceSet::iterator k : chains[c])
                segMap.erase(k);
}

// MappedSegment::canMergeRightWith (api/impl/halMappedSegment.cpp:109-161) with both cut sets
bool canMergeRight(const Piece &a, const Piece &b, const std::set<int64_t> &cutSet, const std::set<int64_t> &sourceCutSet) {
    if (a.tRev != b.tRev || a.sRev != b.sRev)
        return false;
    int64_t qdelta, rdelta, cut, sourceCut;
    if (!a.tRev && !a.sRev) {
        qdelta = b.tStart() - a.tEnd();
        rdelta = b.sStart() - a.sEnd();
        cut = a.tEnd();
        sourceCut = a.sEnd();
    } else if (a.tRev && a.sRev) {
        qdelta = b.tEnd() - a.tStart();
        rdelta = b.sEnd() - a.sStart();
        cut = a.tStart();
        sourceCut = a.sStart();
    } else if (!a.tRev && a.sRev) {
        qdelta = b.tStart() - a.tEnd();
        rdelta = a.sEnd() - b.sStart();
        cut = a.tEnd();
        sourceCut = b.sStart();
    } else {
        qdelta = b.tEnd() - a.tStart();
        rdelta = a.sStart() - b.sEnd();
        cut = a.tStart();
        sourceCut = b.sEnd();
    }
    if (qdelta != 1 || rdelta != 1)
        return false;
    return !sourceCutSet.count(sourceCut) && !cutSet.count(cut);
}

// BlockMapper::extractSegment (liftover/impl/halBlockMapper.cpp:331-394): the fragments that merge to the right of `start`
// leave the set; first and last of them come back
void extractSegment(const Mapper &M, PieceSet::iterator start, const PieceSet &paraSet, PieceSet &set, const std::set<int64_t> &targetCutPoints,
                    std::set<int64_t> &queryCutPoints, Piece &first, Piece &last) {
    first = last = *start;
    const int startSeq = M.Q->seqIndexBySite(start->tLo);
    std::vector<PieceSet::iterator> v1, v2, toErase;
    v1.push_back(start);
    PieceSet::iterator next = start;
    ++next;
    while (next != set.end() && v1.back()->tLo == next->tLo) { // equalTargetStart (halBlockMapper.h:85-92)
        v1.push_back(next);
        ++next;
    }
    while (next != set.end()) {
        while (next != set.end() && (v2.empty() || v2.back()->tLo == next->tLo) && v2.size() < v1.size()) {
            v2.push_back(next);
            ++next;
        }
        bool canMerge = v1.size() == v2.size();
        for (size_t i = 0; i < v1.size() && canMerge; ++i)
            canMerge = M.Q->seqIndexBySite(v2[i]->tLo) == startSeq && canMergeRight(*v1[i], *v2[i], queryCutPoints, targetCutPoints) &&
                       (paraSet.find(*v1[i]) == paraSet.end()) == (paraSet.find(*v2[i]) == paraSet.end());
        if (!canMerge)
            break;
        last = *v2[0];
        toErase.push_back(v2[0]);
        v1.clear();
        std::swap(v1, v2);
    }
    if (v1.size() > 1)
        queryCutPoints.insert(std::max(last.tStart(), last.tEnd()));
    for (PieceSet::iterator e : toErase)
        set.erase(e);
}

char *dupString(const std::string &s) {
    char *p = (char *)malloc(s.size() + 1);
    if (!p)
        throw std::bad_alloc();
    memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

char complement(char c) { // api/impl/halCommon.cpp: reverseComplement(char)
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
}

// Sequence::getSubString of [genomePos, genomePos + n) as a malloc'd C string, reverse-complemented on request
char *dnaString(const GenomeTables &G, int64_t genomePos, int64_t n, bool reverseComplement) {
    char *p = (char *)malloc((size_t)n + 1);
    if (!p)
        throw std::bad_alloc();
    for (int64_t k = 0; k < n; ++k) {
        const char c = G.dna.empty() ? 'N' : dnaAt(G.dna, genomePos + k);
        if (reverseComplement)
            p[n - 1 - k] = complement(c);
        else
            p[k] = c;
    }
    p[n] = '\0';
    return p;
}

// processTargetDupes (halBlockViz.cpp:944-1052)
hgx_target_dupe_list *processTargetDupes(const Mapper &M, const PieceSet &paraSet) {
    std::vector<std::pair<std::set<int64_t>, int64_t>> lists;
    for (PieceSet::const_iterator i = paraSet.begin(); i != paraSet.end();) {
        PieceSet::const_iterator j = i;
        ++j;
        while (j != paraSet.end() && (j->tStart() == i->tStart() || j->tEnd() == i->tStart()))
            ++j;
        std::set<int64_t> starts;
        for (PieceSet::const_iterator k = i; k != j; ++k)
            starts.insert(k->sStart());
        lists.emplace_back(starts, i->length());
        i = j;
    }
    std::sort(lists.begin(), lists.end(), [](const std::pair<std::set<int64_t>, int64_t> &a, const std::pair<std::set<int64_t>, int64_t> &b) {
        return *a.first.begin() < *b.first.begin();
    });
    for (size_t i = 0; i < lists.size(); ++i) {
        if (lists[i].second <= 0)
            continue;
        for (size_t j = i + 1; j < lists.size(); ++j) {
            bool merged = false;
            if (lists[j].first.size() == lists[i].first.size()) {
                auto k1 = lists[i].first.begin(), k2 = lists[j].first.begin();
                int64_t minExtension = std::numeric_limits<int64_t>::max();
                for (; k1 != lists[i].first.end(); ++k1, ++k2) {
                    int64_t leftOverlap = -1;
                    if (*k2 >= *k1) {
                        leftOverlap = (*k1 + lists[i].second) - *k2;
                        if (leftOverlap > 0)
                            leftOverlap = std::min(leftOverlap, lists[j].second);
                    }
                    minExtension = std::min(minExtension, leftOverlap < 0 ? (int64_t)-1 : leftOverlap - lists[j].second);
                }
                if (minExtension == 0) {
                    lists[j].second = 0;
                } else if (minExtension > 0) {
                    lists[i].second += minExtension;
                    lists[j].second -= minExtension;
                }
                merged = minExtension >= 0;
            }
            if (!merged)
                break;
        }
    }
    const SeqInfo &chrom = M.R->seqs[(size_t)M.R->seqIndexBySite(paraSet.begin()->sLo)];
    hgx_target_dupe_list *head = nullptr, *tail = nullptr;
    int64_t curId = 0, prev = -1;
    for (size_t i = 0; i < lists.size(); ++i) {
        if (lists[i].second == 0)
            continue;
        hgx_target_dupe_list *d = (hgx_target_dupe_list *)calloc(1, sizeof *d);
        if (!d)
            throw std::bad_alloc();
        if (head == nullptr)
            head = d;
        else
            tail->next = d;
        tail = d;
        if (prev >= 0 && *lists[i].first.begin() > *lists[(size_t)prev].first.begin() + lists[(size_t)prev].second)
            ++curId;
        d->id = curId;
        d->qChrom = dupString(chrom.name);
        hgx_target_range *rt = nullptr;
        for (int64_t s : lists[i].first) {
            hgx_target_range *r = (hgx_target_range *)calloc(1, sizeof *r);
            if (!r)
                throw std::bad_alloc();
            r->tStart = s - chrom.start;
            r->size = lists[i].second;
            if (rt == nullptr)
                d->tRange = r;
            else
                rt->next = r;
            rt = r;
        }
        prev = (int64_t)i;
    }
    return head;
}

} // namespace

// every range of a call: (absFirst, absLast) in the reference genome, inclusive; results[k] for range k
void blocksInTargetRanges(hgx_alignment *h, int qGenome, int tGenome, const std::vector<std::pair<int64_t, int64_t>> &ranges, bool tReversed,
                          bool getSequenceString, bool doDupes, bool doTargetDupes, bool doAdjes, int coalescenceLimit,
                          std::vector<hgx_block_results *> &results) {
    const Image &img = h->img;
    Mapper M;
    M.h = h;
    M.ref = tGenome;
    M.query = qGenome;
    M.mrca = img.lca(tGenome, qGenome);
    // readBlocks (halBlockViz.cpp:765-788): a self-alignment walks back to the root for its paralogies unless told otherwise
    M.limit = coalescenceLimit >= 0 ? coalescenceLimit : (qGenome == tGenome ? img.root() : M.mrca);
    M.doDupes = doDupes;
    M.R = &img.genomes[(size_t)tGenome];
    M.Q = &img.genomes[(size_t)qGenome];
    // which tiling the mapped segments' own side lies on: the query genome's bottom segments only when it is the MRCA, reached
    // by the walk up alone (halSegmentMapper.cpp:578-637: no paralogy phase, no walk down), its top segments otherwise
    const bool paralogyPhase = M.limit != M.mrca && doDupes;
    M.queryTop = !(qGenome == M.mrca && tGenome != M.mrca && !paralogyPhase);
    results.assign(ranges.size(), nullptr);
    if (ranges.empty())
        return;

    // ---- GPU batch 1: BlockMapper::map for every range ----
    hgx_liftover_opts fwd{};
    fwd.traverse_dupes = doDupes ? 1 : 0;
    fwd.coalescence_limit = M.limit == M.mrca ? -1 : M.limit;
    fwd.min_length = 0;
    fwd.emit_blocks = 1;
    fwd.block_mapper_source = 1;
    std::vector<int64_t> gs, ge;
    std::vector<uint8_t> st;
    for (const auto &r : ranges) {
        gs.push_back(r.first);
        ge.push_back(r.second);
        st.push_back((uint8_t)(tReversed ? '-' : '+'));
    }
    std::vector<hgx_record> recs;
    liftoverBatchAbsolute(h, tGenome, qGenome, gs, ge, st, fwd, recs);
    std::vector<RangeJob> jobs(ranges.size());
    for (size_t k = 0; k < ranges.size(); ++k) {
        jobs[k].absFirst = ranges[k].first;
        jobs[k].absLast = ranges[k].second;
    }
    for (const hgx_record &r : recs)
        jobs[(size_t)r.query].segSet.insert(pieceOf(r, *M.Q));

    // ---- GPU batch 2: the stretches next to every member, mapped back ----
    if (doAdjes) {
        struct Ask {
            size_t job;
            Piece member;
            int side;
        };
        std::vector<Ask> asks;
        gs.clear();
        ge.clear();
        st.clear();
        for (size_t k = 0; k < jobs.size(); ++k)
            for (const Piece &m : jobs[k].segSet)
                for (int side = 0; side < 2; ++side) {
                    int64_t lo, hi, segIdx;
                    if (!neighbour(M, m, side == 0, lo, hi, segIdx))
                        continue;
                    asks.push_back({k, m, side});
                    gs.push_back(lo);
                    ge.push_back(hi);
                    st.push_back((uint8_t)(m.tRev ? '-' : '+')); // (the iterator keeps the member's orientation)
                }
        hgx_liftover_opts bwd{};
        bwd.traverse_dupes = doDupes ? 1 : 0;
        bwd.coalescence_limit = -1; // halMapSegment's defaults (halBlockMapper.cpp:151, :175)
        bwd.min_length = 0;
        bwd.emit_blocks = 2;
        bwd.block_mapper_source = M.queryTop ? 2 : 3;
        liftoverBatchAbsolute(h, qGenome, tGenome, gs, ge, st, bwd, recs);
        // the records of ask a: query == a, in order
        std::vector<size_t> firstRec(asks.size() + 1, recs.size());
        {
            size_t at = 0;
            for (size_t a = 0; a <= asks.size(); ++a) {
                while (at < recs.size() && (size_t)recs[at].query < a)
                    ++at;
                firstRec[a] = at;
            }
        }
        // ---- the sequential part, range by range: the set grows while it is walked (halBlockMapper.cpp:110-118) ----
        struct ByMember {
            bool operator()(const std::pair<Piece, int> &a, const std::pair<Piece, int> &b) const {
                if (PieceLess()(a.first, b.first))
                    return true;
                if (PieceLess()(b.first, a.first))
                    return false;
                return a.second < b.second;
            }
        };
        size_t a0 = 0;
        for (size_t k = 0; k < jobs.size(); ++k) {
            RangeJob &J = jobs[k];
            std::map<std::pair<Piece, int>, size_t, ByMember> askOf;
            for (; a0 < asks.size() && asks[a0].job == k; ++a0)
                askOf[{asks[a0].member, asks[a0].side}] = a0;
            const int refSequence = M.R->seqIndexBySite(J.absFirst);
            for (PieceSet::const_iterator i = J.segSet.begin(); i != J.segSet.end(); ++i) {
                if (J.adjSet.find(*i) != J.adjSet.end())
                    continue;
                std::vector<Piece> back[2];
                for (int side = 0; side < 2; ++side) {
                    auto it = askOf.find({*i, side});
                    if (it == askOf.end())
                        continue;
                    for (size_t r = firstRec[it->second]; r < firstRec[it->second + 1]; ++r)
                        back[side].push_back(pieceOf(recs[r], *M.R));
                }
                mapAdjacencies(M, J, i, back, refSequence);
            }
        }
    }

    // ---- readBlocks: chaining, fragments, blocks, target dupes ----
    for (size_t k = 0; k < jobs.size(); ++k) {
        RangeJob &J = jobs[k];
        PieceSet paraSet;
        if (doDupes && qGenome != tGenome)
            chainReferenceParalogies(J.segSet, paraSet);
        std::set<int64_t> queryCutSet, targetCutSet;
        targetCutSet.insert(J.absFirst);
        targetCutSet.insert(J.absLast);
        hgx_block_results *res = (hgx_block_results *)calloc(1, sizeof *res);
        if (!res)
            throw std::bad_alloc();
        results[k] = res;
        hgx_block *prev = nullptr;
        const std::string qGenomeName = M.Q->name;
        for (PieceSet::iterator it = J.segSet.begin(); it != J.segSet.end(); ++it) {
            Piece first, last;
            extractSegment(M, it, paraSet, J.segSet, targetCutSet, queryCutSet, first, last);
            hgx_block *cur = (hgx_block *)calloc(1, sizeof *cur);
            if (!cur)
                throw std::bad_alloc();
            if (res->mappedBlocks == nullptr)
                res->mappedBlocks = cur;
            else
                prev->next = cur;
            prev = cur;
            // readBlock (halBlockViz.cpp:832-905)
            const SeqInfo &qSeq = M.Q->seqs[(size_t)M.Q->seqIndexBySite(first.tLo)];
            const SeqInfo &tSeq = M.R->seqs[(size_t)M.R->seqIndexBySite(first.sLo)];
            const size_t prefix = qSeq.name.find(qGenomeName + '.') != 0 ? 0 : qGenomeName.length() + 1;
            cur->qChrom = dupString(qSeq.name.substr(prefix));
            cur->tStart = std::min(first.sLo, last.sLo) - tSeq.start;
            cur->qStart = std::min(first.tLo, last.tLo) - qSeq.start;
            const int64_t tEnd = std::max(first.sHi, last.sHi) - tSeq.start;
            cur->size = 1 + tEnd - cur->tStart;
            cur->strand = first.tRev ? '-' : '+';
            if (getSequenceString) {
                cur->qSequence = dnaString(*M.Q, qSeq.start + cur->qStart, cur->size, cur->strand == '-');
                cur->tSequence = dnaString(*M.R, tSeq.start + cur->tStart, cur->size, false);
            }
        }
        if (!paraSet.empty() && doTargetDupes)
            res->targetDupeBlocks = processTargetDupes(M, paraSet);
    }
}

} // namespace hgx
