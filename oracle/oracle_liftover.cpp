// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).
// Restatement of the halLiftover driver:
//   liftover/impl/halBedLine.cpp, halBedScanner.cpp, halLiftover.cpp,
//   halBlockLiftover.cpp, halBlockMapper.cpp (extractSegment).
// BED columns <= 9; BED12 / PSL are §8(f) "next".
#include "oracle_liftover.hpp"
#include <chrono>
#include <iostream>
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
    if (_bedType > 9)
        throw std::runtime_error("oracle: BED12 path not restated yet (SURVEY 8(f))");
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
    for (size_t i = 0; i < _extra.size(); ++i)
        os << '\t' << _extra[i];
    os << '\n';
    return os;
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
    for (MSegSet::iterator i = mappedSegments.begin(); i != mappedSegments.end(); ++i) {
        extractSegment(i, emptySet, fragments, &mappedSegments, targetCutSet, queryCutSet);
        const Sequence *seq = (*i)->tgt.getSequence();
        i64 seqStart = seq->start;
        mappedBedLines.push_back(bedLine);
        BedLine &out = mappedBedLines.back();
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
    }
}

// liftover/impl/halLiftover.cpp:313-355 (bedType <= 9 part)
void Liftover::cleanResults() {
    if (bedLine._bedType > 6) {
        for (auto &b : outBedLines) {
            if (bedLine._thickStart != 0 || bedLine._thickEnd != 0) {
                b._thickStart = b._start;
                b._thickEnd = b._end;
            }
        }
    }
}

// liftover/impl/halLiftover.cpp:46-92
void Liftover::visitLine() {
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
    }
    auto t0 = std::chrono::steady_clock::now();
    mappedBlocks.clear();
    liftInterval(mappedBlocks);
    outBedLines = mappedBlocks; // writeBlocksAsIntervals, halLiftover.cpp:169-171
    cleanResults();
    outBedLines.sort([](const BedLine &a, const BedLine &b) { return a._srcStart < b._srcStart; }); // BedLineSrcLess :202
    mapSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++numIntervals;
    numRecords += outBedLines.size();
    for (auto &b : outBedLines)
        b.write(*outStream);
}

// liftover/impl/halBedScanner.cpp:40-61, :76-80; liftover/impl/halLiftover.cpp:23-41
void Liftover::convert(const Alignment *alignment, int src, std::istream *bedIn, int tgt, std::ostream *bedOut, int bedType,
                       bool doDupes, int coalLimit) {
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

} // namespace orc
