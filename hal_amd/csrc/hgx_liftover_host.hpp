// Host adapter with the reference's Liftover interface (liftover/inc/halLiftover.h:25-28): BED text
// in, BED text out; the per-interval block mapping runs on the GPU through the liftover engine.
#pragma once
#include "hgx_liftover_engine.hpp"
#include <istream>
#include <ostream>
#include <set>

namespace hgx {

// One BED line: liftover/inc/halBedLine.h:48-82 (blocks / PSL fields belong to the BED12 path, not built yet)
struct BedLine {
    std::string chrName;
    int64_t start = NULL_INDEX, end = NULL_INDEX;
    std::string name;
    int64_t score = 0;
    char strand = '+'; // halBedLine.cpp:19
    int64_t thickStart = 0, thickEnd = 0, itemR = 0, itemG = 0, itemB = 0;
    std::vector<std::string> extra;
    int bedType = -1;
    // halBedLine.cpp:27-102; throws std::runtime_error with the reference's messages
    void parse(const std::string &lineBuffer, int bedType);
    // halBedLine.cpp:104-151
    void write(std::ostream &os) const;
};

class Liftover {
  public:
    // Same argument meaning as Liftover::convert (halLiftover.cpp:23-41); genomes are ids of `alignment`.
    // Throws std::runtime_error where the reference throws hal_exception (message + " in input bed line N").
    void convert(hgx_alignment *alignment, int srcGenome, std::istream *inBedStream, int tgtGenome, std::ostream *outBedStream,
                 int bedType = 0, bool traverseDupes = true, bool outPSL = false, bool outPSLWithName = false,
                 int coalescenceLimit = -1);
    // intervals per device batch (memory bound only)
    size_t batchLines = 1u << 22;
    hgx_liftover_stats lastStats{};

  private:
    std::set<std::string> _missedSet;
};

} // namespace hgx
