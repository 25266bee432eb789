// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).
// Restatement of the halLiftover driver:
//   liftover/impl/halBedLine.cpp, halBedScanner.cpp, halLiftover.cpp,
//   halBlockLiftover.cpp, halBlockMapper.cpp (extractSegment).
// BED columns <= 9; BED12 / PSL are §8(f) "next".
#include "oracle_liftover.hpp"
#include <chrono>
#include <iostream>
#include <limits>
#include <sstream>

namespace orc {

// api/impl/halCommon.cpp:28-42
static std::vector<std::string> chopString(const std::string &in, const std::string &sep) {
    std::vector<std::string> out;
    std::string::size_type start = 0, end = 0;
    while ((end = in.find(sep, start)) != std::string::npos) {
        out.push_back(in.substr(start, end - start));
        start = end + sep.size();
    }
    if (start < in.length())
        out.push_back(in.substr(start, std::string::npos));
    return out;
}
// api/impl/halCommon.cpp:44-52
static i64 strToInt(const std::string &s) {
    std::stringstream ss(s);
    i64 i;
    ss >> i;
    if (ss.bad() || ss.fail())
        throw std::runtime_error("Error converting string to int: " + s);
    return i;
}

// liftover/impl/halBedLine.cpp:27-102
std::istream &BedLine::read(std::istream &is, std::string &lineBuffer, int bedType) {
    _bedType = bedType;
    std::getline(is, lineBuffer);
    std::vector<std::string> row = chopString(lineBuffer, "\t");
    if (row.size() < 3)
        throw std::runtime_error("Expected at least three columns in BED record: " + lineBuffer);
    if (_bedType == 0)
        _bedType = std::min(int(row.size()), 12);
    _chrName = row[0];
    _start = strToInt(row[1]);
    _end = strToInt(row[2]);
    if (_start >= _end)
        throw std::runtime_error("Error zero or negative length BED range: " + lineBuffer);
    if (_bedType > 3)
        _name = row[3];
    if (_bedType > 4)
        _score = strToInt(row[4]);
    if (_bedType > 5) {
        _strand = row[5][0];
        if (_strand != '.' && _strand != '+' && _strand != '-')
            throw std::runtime_error("Strand character must be + or - or ." + lineBuffer);
    }
    if (_bedType > 6)
        _thickStart = strToInt(row[6]);
    if (_bedType > 7)
        _thickEnd = strToInt(row[7]);
    if (_bedType > 8) {
        std::vector<std::string> rgb = chopString(row[8], ",");
        if (rgb.size() > 3 || rgb.size() == 0)
            throw std::runtime_error("Error parsing BED itemRGB: " + lineBuffer);
        _itemR = strToInt(rgb[0]);
        _itemG = _itemB = _itemR;
        if (rgb.size() > 1)
            _itemG = strToInt(rgb[1]);
        if (rgb.size() == 3)
            _itemB = strToInt(rgb[2]);
    }
    if (_bedType > 9) {
        if (_bedType < 12)
            throw std::runtime_error("Error parsing BED, insufficient columns for blocks: " + lineBuffer);
        size_t numBlocks = (size_t)strToInt(row[9]);
        std::vector<std::string> blockSizes = chopString(row[10], ",");
        if (blockSizes.size() != numBlocks)
            throw std::runtime_error("Error parsing BED blockSizes: " + lineBuffer);
        std::vector<std::string> blockStarts = chopString(row[11], ",");
        if (blockStarts.size() != numBlocks)
            throw std::runtime_error("Error parsing BED blockStarts: " + lineBuffer);
        _blocks.resize(numBlocks);
        for (size_t i = 0; i < numBlocks; ++i) {
            _blocks[i]._length = strToInt(blockSizes[i]);
            _blocks[i]._start = strToInt(blockStarts[i]);
            if (_start + _blocks[i]._start + _blocks[i]._length > _end)
                throw std::runtime_error("Error BED block out of range: " + lineBuffer);
        }
    }
    _extra.clear();
    for (size_t i = (size_t)_bedType; i < row.size(); i++)
        _extra.push_back(row[i]);
    return is;
}

// liftover/impl/halBedLine.cpp:104-151
std::ostream &BedLine::write(std::ostream &os) const {
    os << _chrName << '\t' << _start << '\t' << _end;
    if (_bedType > 3)
        os << '\t' << _name;
    if (_bedType > 4)
        os << '\t' << _score;
    if (_bedType > 5)
        os << '\t' << _strand;
    if (_bedType > 6)
        os << '\t' << _thickStart;
    if (_bedType > 7)
        os << '\t' << _thickEnd;
    if (_bedType > 8)
        os << '\t' << _itemR << ',' << _itemG << ',' << _itemB;
    if (_bedType > 9) {
        os << '\t' << _blocks.size();
        for (size_t i = 0; i < _blocks.size(); ++i)
            os << (i == 0 ? '\t' : ',') << _blocks[i]._length;
        for (size_t i = 0; i < _blocks.size(); ++i)
            os << (i == 0 ? '\t' : ',') << _blocks[i]._start;
    }
    for (size_t i = 0; i < _extra.size(); ++i)
        os << '\t' << _extra[i];
    os << '\n';
    return os;
}

// liftover/impl/halBedLine.cpp:153-178
void BedLine::expandToBed12() {
    if (_bedType <= 3)
        _name = "";
    if (_bedType <= 4)
        _score = 0;
    if (_bedType <= 5)
        _strand = '+';
    if (_bedType <= 6)
        _thickStart = _start;
    if (_bedType <= 7)
        _thickEnd = _end;
    if (_bedType <= 8)
        _itemR = _itemG = _itemB = 0;
    if (_bedType <= 9) {
        _blocks.resize(1);
        _blocks[0]._start = 0;
        _blocks[0]._length = _end - _start;
    }
    _bedType = 12;
}

// liftover/impl/halBedLine.cpp:206-250
std::ostream &BedLine::writePSL(std::ostream &os, bool prefixWithName) const {
    const PSLInfo &psl = _psl[0];
    if (!validatePSL())
        throw std::runtime_error("Internal error: PSL does not validate");
    if (prefixWithName)
        os << _name << '\t';
    os << psl._matches << '\t' << psl._misMatches << '\t' << psl._repMatches << '\t' << psl._nCount << '\t' << psl._qNumInsert
       << '\t' << psl._qBaseInsert << '\t' << psl._tNumInsert << '\t' << psl._tBaseInsert << '\t' << psl._qStrand << _strand
       << '\t' << psl._qSeqName << '\t' << psl._qSeqSize << '\t' << (_srcStart - (i64)psl._qChromOffset) << '\t'
       << (psl._qEnd - psl._qChromOffset) << '\t' << _chrName << '\t' << psl._tSeqSize << '\t' << _start << '\t' << _end << '\t'
       << _blocks.size() << '\t';
    for (size_t i = 0; i < _blocks.size(); ++i)
        os << _blocks[i]._length << ',';
    os << '\t';
    for (size_t i = 0; i < psl._qBlockStarts.size(); ++i) {
        i64 start = psl._qBlockStarts[i] - (i64)psl._qChromOffset;
        if (psl._qStrand == '-')
            start = (i64)psl._qSeqSize - start - _blocks[i]._length;
        os << start << ',';
    }
    os << '\t';
    for (size_t i = 0; i < _blocks.size(); ++i) {
        i64 start = _blocks[i]._start + _start;
        if (_strand == '-')
            start = (i64)psl._tSeqSize - start - _blocks[i]._length;
        os << start << ',';
    }
    os << '\n';
    return os;
}

// liftover/impl/halBedLine.cpp:252-334
bool BedLine::validatePSL() const {
    if (_psl.size() != 1 || _blocks.size() < 1)
        return false;
    const PSLInfo &psl = _psl[0];
    if (_blocks.size() != psl._qBlockStarts.size())
        return false;
    u64 totBlockLen = 0;
    for (size_t i = 0; i < _blocks.size(); ++i)
        totBlockLen += (u64)_blocks[i]._length;
    if (totBlockLen != psl._matches + psl._misMatches + psl._repMatches + psl._nCount)
        return false;
    if (totBlockLen + psl._qBaseInsert != psl._qEnd - (u64)_srcStart)
        return false;
    if (totBlockLen + psl._tBaseInsert != (u64)_end - (u64)_start)
        return false;
    if (_strand != '-') {
        if (_blocks[0]._start != 0 || _blocks.back()._start + _blocks.back()._length + _start != _end)
            return false;
    } else {
        if (_blocks.back()._start != 0 || _blocks[0]._start + _blocks[0]._length + _start != _end)
            return false;
    }
    if (psl._qStrand != '-') {
        if (psl._qBlockStarts[0] != _srcStart || (u64)(psl._qBlockStarts.back() + _blocks.back()._length) != psl._qEnd)
            return false;
    } else {
        if (psl._qBlockStarts.back() != _srcStart || (u64)(psl._qBlockStarts[0] + _blocks[0]._length) != psl._qEnd)
            return false;
    }
    return true;
}

// liftover/inc/halBlockMapper.h:85-92
static bool equalTargetStart(const MSegPtr &s1, const MSegPtr &s2) {
    i64 p1 = std::min(s1->getStartPosition(), s1->getEndPosition());
    i64 p2 = std::min(s2->getStartPosition(), s2->getEndPosition());
    return p1 == p2;
}

// liftover/impl/halBlockMapper.cpp:331-394
void extractSegment(MSegSet::iterator start, const MSegSet &paraSet, std::vector<MSegPtr> &fragments, MSegSet *startSet,
                    const std::set<i64> &targetCutPoints, std::set<i64> &queryCutPoints) {
    fragments.clear();
    fragments.push_back(*start);
    const Sequence *startSeq = (*start)->tgt.getSequence();
    std::vector<MSegSet::iterator> vector1, vector2, toErase;
    std::vector<MSegSet::iterator> *v1 = &vector1, *v2 = &vector2;
    v1->push_back(start);
    MSegSet::iterator next = start;
    ++next;
    while (next != startSet->end() && equalTargetStart(*v1->back(), *next)) {
        v1->push_back(next);
        ++next;
    }
    while (next != startSet->end()) {
        while (next != startSet->end() && (v2->empty() || equalTargetStart(*v2->back(), *next)) && v2->size() < v1->size()) {
            v2->push_back(next);
            ++next;
        }
        bool canMerge = v1->size() == v2->size();
        for (size_t i = 0; i < v1->size() && canMerge; ++i) {
            canMerge = (v1->size() == v2->size() && (*v2->at(i))->tgt.getSequence() == startSeq &&
                        canMergeRightWith(**v1->at(i), **v2->at(i), &queryCutPoints, &targetCutPoints) &&
                        (paraSet.find(*v1->at(i)) == paraSet.end()) == (paraSet.find(*v2->at(i)) == paraSet.end()));
        }
        if (canMerge) {
            fragments.push_back(*v2->at(0));
            toErase.push_back(v2->at(0));
        } else {
            break;
        }
        v1->clear();
        std::swap(v1, v2);
    }
    if (v1->size() > 1)
        queryCutPoints.insert(std::max(fragments.back()->getStartPosition(), fragments.back()->getEndPosition()));
    for (size_t i = 0; i < toErase.size(); ++i)
        startSet->erase(toErase[i]);
}

// liftover/impl/halBlockLiftover.cpp:23-44
void Liftover::visitBegin() {
    const Genome &S = al->genomes[(size_t)srcGenome];
    refSeg = SegIt();
    refSeg.al = al;
    refSeg.g = srcGenome;
    if (S.numTop > 0) {
        refSeg.top = true;
        lastIndex = S.numTop;
    } else {
        refSeg.top = false;
        lastIndex = S.numBot;
    }
    std::set<int> in;
    in.insert(srcGenome);
    in.insert(tgtGenome);
    mrca = getLowestCommonAncestor(*al, in);
    if (coalescenceLimit < 0)
        coalescenceLimit = mrca;
    in.clear();
    in.insert(coalescenceLimit);
    in.insert(tgtGenome);
    downwardPath.clear();
    getGenomesInSpanningTree(*al, in, downwardPath);
}

// liftover/impl/halBlockLiftover.cpp:46-113
void Liftover::liftInterval(std::list<BedLine> &mappedBedLines) {
    mappedSegments.clear();
    i64 globalStart = bedLine._start + srcSequence->start;
    i64 globalEnd = bedLine._end - 1 + srcSequence->start;
    bool flip = bedLine._strand == '-';
    refSeg.rev = false;
    refSeg.toSite(globalStart, false);
    i64 startOffset = globalStart - refSeg.getStartPosition();
    i64 endOffset = 0;
    if (globalEnd <= refSeg.getEndPosition())
        endOffset = refSeg.getEndPosition() - globalEnd;
    refSeg.slice(startOffset, endOffset);
    while (refSeg.idx < lastIndex && refSeg.getStartPosition() <= globalEnd) {
        if (flip)
            refSeg.toReverseInPlace();
        halMapSegment(refSeg, mappedSegments, tgtGenome, &downwardPath, traverseDupes, 0, coalescenceLimit, mrca);
        if (flip)
            refSeg.toReverseInPlace();
        refSeg.toRight(globalEnd);
    }
    numMappedPieces += mappedSegments.size();
    std::vector<MSegPtr> fragments;
    MSegSet emptySet;
    std::set<i64> queryCutSet, targetCutSet;
    struct Record { // hgx_record
        i64 query, tgtStart, tgtEnd, srcStart;
        int32_t tgtSeq;
        char strand;
        u8 tgtReversed;
        char pad[2];
    };
    static_assert(sizeof(Record) == 40, "hgx_record");
    std::vector<Record> records;
    for (MSegSet::iterator i = mappedSegments.begin(); i != mappedSegments.end(); ++i) {
        extractSegment(i, emptySet, fragments, &mappedSegments, targetCutSet, queryCutSet);
        const Sequence *seq = (*i)->tgt.getSequence();
        i64 seqStart = seq->start;
        mappedBedLines.push_back(bedLine);
        BedLine &out = mappedBedLines.back();
        out._blocks.clear();
        out._chrName = seq->name;
        out._start = std::min(std::min(fragments.front()->getStartPosition(), fragments.front()->getEndPosition()),
                              std::min(fragments.back()->getStartPosition(), fragments.back()->getEndPosition()));
        out._start -= seqStart;
        out._end = 1 + std::max(std::max(fragments.front()->getStartPosition(), fragments.front()->getEndPosition()),
                                std::max(fragments.back()->getStartPosition(), fragments.back()->getEndPosition()));
        out._end -= seqStart;
        out._strand = (*i)->getReversed() ? '-' : '+';
        const SegIt &srcFront = fragments.front()->src;
        const SegIt &srcBack = fragments.back()->src;
        out._srcStart = std::min(std::min(srcFront.getStartPosition(), srcFront.getEndPosition()),
                                 std::min(srcBack.getStartPosition(), srcBack.getEndPosition()));
        out._srcStrand = srcFront.rev ? '-' : '+';
        if (bedLine._strand == '.') {
            out._strand = '.';
            out._srcStrand = '.';
        }
        if (outPSL && !fragments.empty())
            readPSLInfo(fragments, out);
        if (recordsOut)
            records.push_back(Record{recordQuery, out._start, out._end, out._srcStart, (int32_t)(seq - (*i)->tgt.G().seqs.data()), out._strand,
                                     (u8)((*i)->getReversed() ? 1 : 0), {0, 0}});
    }
    if (recordsOut) {
        std::stable_sort(records.begin(), records.end(), [](const Record &a, const Record &b) { return a.srcStart < b.srcStart; });
        recordsOut->write((const char *)records.data(), (std::streamsize)(records.size() * sizeof(Record)));
        ++recordQuery;
    }
}

// DNA of a sliced segment in iteration order (SegmentIterator::getString, halSegmentIterator.cpp:69-76)
static void segString(const SegIt &it, std::string &out) {
    static const char unpack[16] = {'a', 'c', 'g', 't', 'n', 0, 0, 0, 'A', 'C', 'G', 'T', 'N', 0, 0, 0};
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
    const std::vector<u8> &dna = it.G().dna;
    const i64 len = it.getLength();
    out.resize((size_t)len);
    i64 pos = it.getStartPosition();
    for (i64 k = 0; k < len; ++k) {
        const u8 b = dna[(size_t)(pos >> 1)];
        char c = unpack[(pos & 1) ? (b & 0x0F) : (b >> 4)];
        out[(size_t)k] = it.rev ? comp(c) : c;
        pos += it.rev ? -1 : 1;
    }
}

// liftover/impl/halBlockLiftover.cpp:115-162
void Liftover::readPSLInfo(std::vector<MSegPtr> &fragments, BedLine &outBedLine) {
    const Sequence *srcSeq = fragments[0]->src.getSequence();
    const Sequence *tSeq = fragments[0]->tgt.getSequence();
    outBedLine._psl.resize(1);
    PSLInfo &psl = outBedLine._psl[0];
    psl = PSLInfo();
    psl._qSeqName = srcSeq->name;
    psl._qSeqSize = (u64)srcSeq->length;
    psl._qStrand = fragments[0]->src.rev ? '-' : '+';
    psl._qChromOffset = (u64)srcSeq->start;
    psl._qEnd = (u64)(outBedLine._srcStart + (outBedLine._end - outBedLine._start));
    psl._tSeqSize = (u64)tSeq->length;
    std::string sBuf, tBuf;
    for (size_t i = 0; i < fragments.size(); ++i) {
        segString(fragments[i]->src, sBuf);
        segString(fragments[i]->tgt, tBuf);
        for (size_t j = 0; j < sBuf.length(); ++j) {
            if (sBuf[j] == tBuf[j]) {
                if (!(sBuf[j] >= 'a') && !(tBuf[j] >= 'a')) // isMasked, halCommon.h:130-132
                    ++psl._matches;
                else
                    ++psl._repMatches;
            } else if (tBuf[j] == 'n' || tBuf[j] == 'N') { // isMissingData, :126-128
                ++psl._nCount;
            } else {
                ++psl._misMatches;
            }
        }
    }
}

// liftover/impl/halLiftover.cpp:296-309
void Liftover::liftBlockIntervals() {
    BedLine originalBedLine = bedLine;
    std::sort(bedLine._blocks.begin(), bedLine._blocks.end());
    for (auto blockIt = bedLine._blocks.begin(); blockIt != bedLine._blocks.end(); ++blockIt) {
        bedLine._start = blockIt->_start + originalBedLine._start;
        bedLine._end = bedLine._start + blockIt->_length;
        if (bedLine._end > bedLine._start)
            liftInterval(mappedBlocks);
    }
    bedLine._start = originalBedLine._start;
    bedLine._end = originalBedLine._end;
}

// liftover/impl/halLiftover.cpp:108-167
void Liftover::assignBlocksToIntervals() {
    mappedBlocks.sort([](const BedLine &a, const BedLine &b) { return a._srcStart < b._srcStart; });
    i64 prevSrcBlockEnd = NULL_INDEX;
    for (auto blockIt = mappedBlocks.begin(); blockIt != mappedBlocks.end(); ++blockIt) {
        auto blockNext = blockIt;
        ++blockNext;
        i64 srcBlockEnd = blockIt->_srcStart + (blockIt->_end - blockIt->_start);
        bool dupe = (blockIt->_srcStart < prevSrcBlockEnd) || (blockNext != mappedBlocks.end() && blockNext->_srcStart < srcBlockEnd);
        if (outBedLines.empty() || (outPSL && dupe) || !compatible(outBedLines.back(), *blockIt))
            outBedLines.push_back(*blockIt);
        prevSrcBlockEnd = blockIt->_srcStart + (blockIt->_end - blockIt->_start);
        BedLine &tgtBed = outBedLines.back();
        tgtBed._start = std::min(tgtBed._start, blockIt->_start);
        tgtBed._end = std::max(tgtBed._end, blockIt->_end);
        BedBlock block;
        block._start = blockIt->_start;
        block._length = blockIt->_end - blockIt->_start;
        tgtBed._blocks.push_back(block);
        if (outPSL) {
            tgtBed._psl[0]._qBlockStarts.push_back(blockIt->_srcStart);
            if (tgtBed._blocks.size() > 1) {
                tgtBed._psl[0]._matches += blockIt->_psl[0]._matches;
                tgtBed._psl[0]._misMatches += blockIt->_psl[0]._misMatches;
                tgtBed._psl[0]._repMatches += blockIt->_psl[0]._repMatches;
                tgtBed._psl[0]._nCount += blockIt->_psl[0]._nCount;
            }
        }
    }
    for (auto &b : outBedLines)
        for (size_t i = 0; i < b._blocks.size(); ++i)
            b._blocks[i]._start -= b._start;
    if (!outBedLines.empty())
        flipBlocks(outBedLines);
    if (outPSL)
        computePSLInserts(outBedLines);
}

// liftover/impl/halLiftover.cpp:169-195
bool Liftover::compatible(const BedLine &tgtBed, const BedLine &newBlock) {
    if (tgtBed._strand != newBlock._strand)
        return false;
    if (tgtBed._srcStart == newBlock._srcStart)
        return false;
    i64 delta;
    const BedBlock &tgtBlock = tgtBed._blocks.back();
    if (tgtBed._strand != bedLine._strand)
        delta = tgtBlock._start - newBlock._end;
    else
        delta = newBlock._start - (tgtBlock._start + tgtBlock._length);
    if (delta < 0)
        return false;
    if (tgtBed._chrName != newBlock._chrName)
        return false;
    return true;
}

// liftover/impl/halLiftover.cpp:197-234
void Liftover::flipBlocks(std::list<BedLine> &bedList) {
    for (auto &b : bedList) {
        if (b._blocks.size() > 1) {
            i64 delta = b._blocks[1]._start - (b._blocks[0]._start + b._blocks[0]._length);
            bool mustFlip;
            if (!outPSL)
                mustFlip = delta < 0;
            else
                mustFlip = (b._strand == '-' && delta >= 0) || (b._strand != '-' && delta < 0);
            if (mustFlip) {
                std::reverse(b._blocks.begin(), b._blocks.end());
                if (outPSL)
                    std::reverse(b._psl[0]._qBlockStarts.begin(), b._psl[0]._qBlockStarts.end());
            }
        }
    }
}

// liftover/impl/halLiftover.cpp:236-290
void Liftover::computePSLInserts(std::list<BedLine> &bedList) {
    for (auto &bed : bedList) {
        PSLInfo &psl = bed._psl[0];
        psl._qNumInsert = psl._qBaseInsert = psl._tNumInsert = psl._tBaseInsert = 0;
        auto blockIt = bed._blocks.begin();
        auto blockPrev = blockIt;
        auto qStartIt = psl._qBlockStarts.begin();
        auto qStartPrev = qStartIt;
        if (blockIt != bed._blocks.end()) {
            ++blockIt;
            ++qStartIt;
        }
        for (; blockIt != bed._blocks.end(); ++blockIt, ++blockPrev, ++qStartIt, ++qStartPrev) {
            if (bed._strand == '-')
                std::swap(blockIt, blockPrev);
            u64 gap = (u64)(blockIt->_start - (blockPrev->_start + blockPrev->_length));
            if (gap > 0) {
                ++psl._tNumInsert;
                psl._tBaseInsert += gap;
            }
            if (bed._strand == '-')
                std::swap(blockIt, blockPrev);
            if (psl._qStrand == '-') {
                std::swap(qStartIt, qStartPrev);
                std::swap(blockIt, blockPrev);
            }
            if (*qStartIt >= (*qStartPrev + blockPrev->_length))
                gap = (u64)(*qStartIt - (*qStartPrev + blockPrev->_length));
            else
                gap = 0;
            if (gap > 0) {
                ++psl._qNumInsert;
                psl._qBaseInsert += gap;
            }
            if (psl._qStrand == '-') {
                std::swap(qStartIt, qStartPrev);
                std::swap(blockIt, blockPrev);
            }
        }
    }
}

// liftover/impl/halLiftover.cpp:313-355
void Liftover::cleanResults() {
    if (bedLine._bedType > 6) {
        for (auto i = outBedLines.begin(); i != outBedLines.end();) {
            auto j = i;
            ++j;
            if (bedLine._thickStart != 0 || bedLine._thickEnd != 0) {
                i->_thickStart = i->_start;
                i->_thickEnd = i->_end;
            }
            if (bedLine._bedType > 9) {
                if (i->_blocks.size() > 0) {
                    if (outPSL) {
                        i->_srcStart = std::numeric_limits<i64>::max();
                        i->_psl[0]._qEnd = 0;
                        for (size_t k = 0; k < i->_psl[0]._qBlockStarts.size(); ++k) {
                            i->_srcStart = std::min(i->_srcStart, i->_psl[0]._qBlockStarts[k]);
                            i->_psl[0]._qEnd = std::max(i->_psl[0]._qEnd, (u64)i->_psl[0]._qBlockStarts[k] + (u64)i->_blocks[k]._length);
                        }
                    }
                } else {
                    outBedLines.erase(i);
                }
            }
            i = j;
        }
    }
}

// liftover/impl/halLiftover.cpp:46-92
void Liftover::visitLine() {
    if ((outPSL || outPSLWithName) && bedLine._bedType < 12)
        bedLine.expandToBed12(); // forcing to BED12 makes PSL code simpler (halLiftover.cpp:47-50)
    outBedLines.clear();
    srcSequence = al->genomes[(size_t)srcGenome].seqByName(bedLine._chrName);
    if (srcSequence == nullptr) {
        if (missedSet.insert(bedLine._chrName).second)
            std::cerr << "Unable to find sequence " << bedLine._chrName << " in genome " << al->genomes[(size_t)srcGenome].name
                      << std::endl;
        return;
    } else if (bedLine._end > srcSequence->length) {
        std::cerr << "Skipping interval with endpoint " << bedLine._end << "because sequence " << bedLine._chrName
                  << " has length " << srcSequence->length << std::endl;
        return;
    } else if (bedLine._bedType > 9 && bedLine._blocks.empty()) {
        std::cerr << "Skipping input line with 0 blocks" << std::endl;
        return;
    }
    auto t0 = std::chrono::steady_clock::now();
    mappedBlocks.clear();
    if (bedLine._bedType <= 9)
        liftInterval(mappedBlocks);
    else
        liftBlockIntervals();
    if (mappedBlocks.size() > 0 && bedLine._bedType > 9)
        assignBlocksToIntervals();
    if (bedLine._bedType <= 9)
        outBedLines = mappedBlocks; // writeBlocksAsIntervals, halLiftover.cpp:292-294
    cleanResults();
    outBedLines.sort([](const BedLine &a, const BedLine &b) { return a._srcStart < b._srcStart; }); // BedLineSrcLess :202
    mapSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++numIntervals;
    numRecords += outBedLines.size();
    for (auto &b : outBedLines) { // writeLineResults, halLiftover.cpp:97-106
        if (!outPSL)
            b.write(*outStream);
        else
            b.writePSL(*outStream, outPSLWithName);
    }
}

// liftover/impl/halBedScanner.cpp:40-61, :76-80; liftover/impl/halLiftover.cpp:23-41
void Liftover::convert(const Alignment *alignment, int src, std::istream *bedIn, int tgt, std::ostream *bedOut, int bedType,
                       bool doDupes, int coalLimit, bool psl, bool pslWithName) {
    outPSL = psl || pslWithName; // halLiftoverMain.cpp:82-84
    outPSLWithName = pslWithName;
    al = alignment;
    srcGenome = src;
    tgtGenome = tgt;
    coalescenceLimit = coalLimit;
    outStream = bedOut;
    traverseDupes = doDupes;
    missedSet.clear();
    visitBegin();
    std::string lineBuffer;
    size_t lineNumber = 0;
    auto skipWhiteSpaces = [](std::istream *s) {
        while (s->good() && std::isspace((char)s->peek()))
            s->get();
    };
    try {
        skipWhiteSpaces(bedIn);
        while (bedIn->good()) {
            ++lineNumber;
            bedLine.read(*bedIn, lineBuffer, bedType);
            visitLine();
            skipWhiteSpaces(bedIn);
        }
    } catch (std::runtime_error &e) {
        throw std::runtime_error(std::string(e.what()) + " in input bed line " + std::to_string(lineNumber));
    }
}

// liftover/impl/halBlockMapper.cpp:36-110 (mapTargetAdjacencies == false)
void blockMap(const Alignment &al, int refGenome, int queryGenome, i64 absRefFirst, i64 absRefLast, bool targetReversed, bool doDupes,
              i64 minLength, int coalescenceLimit, MSegSet &segSet) {
    std::set<int> in;
    in.insert(refGenome);
    in.insert(queryGenome);
    const int mrca = getLowestCommonAncestor(al, in);
    if (coalescenceLimit < 0)
        coalescenceLimit = mrca;
    in.clear();
    in.insert(queryGenome);
    in.insert(coalescenceLimit);
    std::set<int> downwardPath;
    getGenomesInSpanningTree(al, in, downwardPath);
    const Genome &R = al.genomes[(size_t)refGenome];
    SegIt refSeg;
    refSeg.al = &al;
    refSeg.g = refGenome;
    i64 lastIndex;
    if (mrca == refGenome && refGenome != queryGenome) { // :79-86
        refSeg.top = false;
        lastIndex = R.numBot;
    } else {
        refSeg.top = true;
        lastIndex = R.numTop;
    }
    if (lastIndex == 0)
        return;
    refSeg.rev = false;
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

} // namespace orc
