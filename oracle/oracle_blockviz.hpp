// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).
//
// Restatement of BlockMapper with adjacencies (liftover/impl/halBlockMapper.cpp:20-330) and of the block reader of the
// blockViz C API on top of it (blockViz/impl/halBlockViz.cpp:243-330, 759-1178: halGetBlocksInTargetRange, readBlocks,
// readBlock, chainReferenceParalogies, processTargetDupes).  Control flow, containers and evaluation order follow the
// reference line by line (sets of shared pointers whose members are flipped in place, iteration over a set that grows).
//
// Pinned on blockViz/tests/expected/blockVizMmapTests.out (the output of blockVizTest --verbose --doSeq on the file of
// blockViz/Makefile:57-71): tests/test_oracle_golden.py.
#pragma once
#include "oracle_liftover.hpp"
#include <ostream>

namespace orc {

// liftover/inc/halBlockMapper.h
struct BlockMapper {
    static i64 maxAdjScan; // halBlockMapper.cpp:20
    const Alignment *al = nullptr;
    MSegSet segSet, adjSet;
    std::set<int> downwardPath, upwardPath;
    int refGenome = -1, queryGenome = -1, mrca = -1, coalescenceLimit = -1;
    const Sequence *refSequence = nullptr;
    i64 absRefFirst = 0, absRefLast = 0, minLength = 0;
    bool targetReversed = false, doDupes = true, mapAdj = false;

    void erase();
    void init(const Alignment *alignment, int refGenome, int queryGenome, i64 absRefFirst, i64 absRefLast, bool targetReversed, bool doDupes,
              i64 minLength, bool mapTargetAdjacencies, int coalescenceLimit = -1);
    void map();
    void mapAdjacencies(MSegSet::const_iterator segIt);
    static SegIt makeIterator(const MSegPtr &mappedSegment, i64 &minIndex, i64 &maxIndex);
    static bool cutByNext(SegIt &query, const SegIt &nextSeg, bool right);
};

// blockViz/inc/halBlockViz.h:32-58, as plain values
struct VizBlock {
    std::string qChrom;
    i64 tStart = 0, qStart = 0, size = 0;
    char strand = '+';
    std::string qSequence, tSequence; // empty when no sequence was asked for
};
struct VizTargetDupe {
    i64 id = 0;
    std::string qChrom;
    std::vector<std::pair<i64, i64>> ranges; // tStart, size
};
struct VizResults {
    std::vector<VizBlock> mappedBlocks;
    std::vector<VizTargetDupe> targetDupeBlocks;
};

// hal_dup_type_t
enum { VIZ_NO_DUPS = 0, VIZ_QUERY_DUPS = 1, VIZ_QUERY_AND_TARGET_DUPS = 2 };

// halGetBlocksInTargetRange (halBlockViz.cpp:243-330) on an open alignment; throws std::runtime_error with the reference's
// messages.  coalescenceLimit: genome index or -1 (= the name was NULL).
VizResults getBlocksInTargetRange(const Alignment &al, int qGenome, int tGenome, const std::string &tChrom, i64 tStart, i64 tEnd,
                                  bool tReversed, bool getSequenceString, int dupMode, bool mapBackAdjacencies, int coalescenceLimit);

// blockVizTest's printBlock / printDupeList (blockViz/tests/blockVizTest.cpp:103-113)
void printVizResults(std::ostream &os, const VizResults &r, bool withSequence);

} // namespace orc
