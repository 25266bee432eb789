// Host adapter with the reference's Liftover interface (liftover/inc/halLiftover.h:25-28): BED text
// in, BED text out; the per-interval block mapping runs on the GPU through the liftover engine.
#pragma once
#include "hgx_liftover_engine.hpp"
#include <istream>
#include <ostream>
#include <set>

namespace hgx {

// liftover/inc/halBedLine.h:19-46
struct BedBlock {
    int64_t start, length;
    bool operator<(const BedBlock &o) const { return start < o.start; }
};
struct PSLInfo {
    uint64_t matches = 0, misMatches = 0, repMatches = 0, nCount = 0, qNumInsert = 0, qBaseInsert = 0, tNumInsert = 0, tBaseInsert = 0;
    std::string qSeqName;
    uint64_t qSeqSize = 0;
    char qStrand = '+';
    uint64_t qEnd = 0, qChromOffset = 0, tSeqSize = 0;
    std::vector<int64_t> qBlockStarts;
};

// One BED line: liftover/inc/halBedLine.h:48-82
struct BedLine {
    std::string chrName;
    int64_t start = NULL_INDEX, end = NULL_INDEX;
    std::string name;
    int64_t score = 0;
    char strand = '+'; // halBedLine.cpp:19
    int64_t thickStart = 0, thickEnd = 0, itemR = 0, itemG = 0, itemB = 0;
    std::vector<BedBlock> blocks;
    std::vector<std::string> extra;
    std::vector<PSLInfo> psl; // at most one element (halBedLine.h:79-81)
    int bedType = -1;
    int64_t srcStart = NULL_INDEX; // hidden sort key (halBedLine.h:74)
    // halBedLine.cpp:27-102; throws std::runtime_error with the reference's messages
    void parse(const std::string &lineBuffer, int bedType);
    // halBedLine.cpp:104-151
    void write(std::ostream &os) const;
    void append(std::string &buf, const std::string &chrom, int64_t s, int64_t e, char str, int64_t tStart, int64_t tEnd) const;
    // halBedLine.cpp:153-178, 206-250, 252-334
    void expandToBed12();
    void writePSL(std::ostream &os, bool prefixWithName) const;
    bool validatePSL() const;
};

class Liftover {
  public:
    // Same argument meaning as Liftover::convert (halLiftover.cpp:23-41); genomes are ids of `alignment`.
    // Throws std::runtime_error where the reference throws hal_exception (message + " in input bed line N").
    void convert(hgx_alignment *alignment, int srcGenome, std::istream *inBedStream, int tgtGenome, std::ostream *outBedStream,
                 int bedType = 0, bool traverseDupes = true, bool outPSL = false, bool outPSLWithName = false,
                 int coalescenceLimit = -1);
    // The same over text in memory; *out receives the lifted text (also what was written before a malformed line, which is
    // then reported by an exception like convert's).  This is what hgx_liftover_convert calls: inputs of up to nine columns,
    // the same number on every line, BED out, take the parallel path of hgx_liftover_text.cpp.
    // *outText: malloc'd, NUL-terminated, owned by the caller (also set when an exception reports a malformed line)
    void convertBuffer(hgx_alignment *alignment, int srcGenome, const char *text, size_t len, int tgtGenome, char **outText, size_t *outLen,
                       int bedType = 0, bool traverseDupes = true, bool outPSL = false, bool outPSLWithName = false, int coalescenceLimit = -1);
    // further handles of the same alignment on other devices (hgx_clone_to_device): convert / convertBuffer shard the lines
    // of inputs the parallel text path takes over `alignment` and these
    std::vector<hgx_alignment *> moreDevices;
    // intervals per device batch (memory bound only; HGX_BATCH_LINES in the environment overrides it: tests)
    size_t batchLines = 1u << 22;
    hgx_liftover_stats lastStats{};

  private:
    // every shape of input: BED12 blocks, PSL output, lines of different column counts (one line at a time on the host)
    void convertGeneral(hgx_alignment *alignment, int srcGenome, std::istream *inBedStream, int tgtGenome, std::ostream *outBedStream, int bedType,
                        bool traverseDupes, bool outPSL, bool outPSLWithName, int coalescenceLimit);
    std::set<std::string> _missedSet;
    bool _outPSL = false, _outPSLWithName = false;
    char _inStrand = '+';
    // halLiftover.cpp:108-290
    void assignBlocksToIntervals(std::vector<BedLine> &mappedBlocks, std::vector<BedLine> &out);
    bool compatible(const BedLine &tgtBed, const BedLine &newBlock) const;
    void flipBlocks(std::vector<BedLine> &lines) const;
    void computePSLInserts(std::vector<BedLine> &lines) const;
};

// hgx_liftover_text.cpp; false: not an input for the fast path (nothing was done)
// als: one handle per device (hgx_clone_to_device); the lines are dealt to them in contiguous shares
// the BED lines of the intervals a writer's wire blobs hold (hgx_liftover_text.cpp)
void liftoverRenderBlobs(hgx_alignment *al, int srcGenome, int tgtGenome, const char *text, size_t len, int bedType, const void *const *blobs,
                         const size_t *blobBytes, int nBlobs, char **outText, size_t *outLen);
bool liftoverTextFast(hgx_alignment *const *als, int nAls, int srcGenome, const char *text, size_t len, int tgtGenome, int bedType,
                      bool traverseDupes, int coalescenceLimit, char **outText, size_t *outLen, std::string &error,
                      std::set<std::string> &missedSet, hgx_liftover_stats &stats, size_t batchLines);

} // namespace hgx
