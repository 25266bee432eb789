// Internal C++ interface of the liftover engine (implemented in hgx_liftover.hip).
#pragma once
#include "../../include/hgx.h"
#include "hgx_device.hpp"
#include <functional>
#include <string>
#include <vector>

namespace hgx {

hgx_liftover_plan *createLiftoverPlan(hgx_alignment *h, int src, int tgt, const hgx_liftover_opts &opts, size_t maxQueries,
                                      bool allowComposed = true, bool tableBuilder = false);
void submitLiftoverPlan(hgx_liftover_plan *p, size_t n, const int64_t *dStart, const int64_t *dEnd, const uint8_t *dStrand, void *stream);
void collectLiftoverPlan(hgx_liftover_plan *p, const hgx_record **dOut, size_t *nOut);
void runLiftoverPlan(hgx_liftover_plan *p, size_t n, const int64_t *dStart, const int64_t *dEnd, const uint8_t *dStrand, void *stream,
                     const hgx_record **dOut, size_t *nOut);
void destroyLiftoverPlan(hgx_liftover_plan *p);
const hgx_liftover_stats &liftoverPlanStats(const hgx_liftover_plan *p);
std::string liftoverPlanKernelTimes(hgx_liftover_plan *p);
// phases of the last table build of this process (HGX_BUILD_TIMING), as a JSON list of [name, ms]
std::string liftoverBuildPhases();
void liftoverPlanSetTiming(hgx_liftover_plan *p, int mode);
void liftoverPlanSetWorkers(hgx_liftover_plan *p, int n);
void liftoverPlanCopyRecords(const hgx_liftover_plan *p, void *dDst, size_t nRecords, void *stream);
void liftoverPlanCopyRecordsPacked(const hgx_liftover_plan *p, void *dDst, size_t nRecords, void *stream);
size_t liftoverPlanWireBlob(hgx_liftover_plan *p, void *dDst, size_t capacity, int64_t firstQuery, int *format, void *stream);
void liftoverBatchHost(hgx_alignment *h, int src, int tgt, size_t n, const hgx_interval *iv, const hgx_liftover_opts &opts,
                       std::vector<hgx_record> &out, hgx_liftover_stats *stats);
// same, the records copied from the device straight into the memory alloc(n) returns
void liftoverBatchHostRaw(hgx_alignment *h, int src, int tgt, size_t n, const hgx_interval *iv, const hgx_liftover_opts &opts,
                          const std::function<hgx_record *(size_t)> &alloc);
// The text path's batch: pinned arrays for n intervals' inclusive genome coordinates and strands (empty interval: start 0,
// end -1) ...
void liftoverStageQueries(hgx_alignment *h, size_t n, int64_t **gs, int64_t **ge, uint8_t **strand);
// ... and the run over what was written into them: *recs (pinned, owned by the alignment, valid until the next staged run)
// packed (optional): a caller that only prints BED lines takes the records in the 8-byte form of the wire (hgx.h: format 8: tgt_start,
// length | tgt_seq << 22 | strand code << 29 | reversed << 31) with the first record of every interval — 30 MB across PCIe for a
// million intervals instead of 130 — whenever the batch fits that form and the plan writes its records densely with their offsets (the
// single-pass path); then *recs is null and packed->words is not.
struct PackedRecords {
    const uint32_t *words = nullptr; // two per record
    const uint32_t *first = nullptr; // n + 1 entries: interval q's records are [first[q], first[q + 1])
};
void liftoverBatchStaged(hgx_alignment *h, int src, int tgt, size_t n, const hgx_liftover_opts &opts, const hgx_record **recs, size_t *nRecs,
                         hgx_liftover_stats *stats, PackedRecords *packed = nullptr);
// the table of the whole path src -> dst with dupes (hgx_device.hpp: ComposedUp, through), built on first use; null when the
// pair cannot have one
const ComposedUp *wholePathTable(hgx_alignment *h, int src, int dst, int coalescenceLimit);
// a batch of absolute intervals (inclusive genome coordinates of src, strand '+' / '-' / '.') through a plan of its own
void liftoverBatchAbsolute(hgx_alignment *h, int src, int tgt, const std::vector<int64_t> &gs, const std::vector<int64_t> &ge,
                           const std::vector<uint8_t> &strand, const hgx_liftover_opts &opts, std::vector<hgx_record> &out);
// BlockMapper::init + map + getMap without adjacencies (liftover/impl/halBlockMapper.cpp:33-110)
void blockMapHost(hgx_alignment *h, int ref, int query, int64_t absFirst, int64_t absLast, bool targetReversed,
                  const hgx_liftover_opts &opts, std::vector<hgx_record> &out);

// hgx_blockviz.cpp: halGetBlocksInTargetRange's readBlocks for every (absolute, inclusive) range of the target genome
void blocksInTargetRanges(hgx_alignment *h, int qGenome, int tGenome, const std::vector<std::pair<int64_t, int64_t>> &ranges, bool tReversed,
                          bool getSequenceString, bool doDupes, bool doTargetDupes, bool doAdjes, int coalescenceLimit,
                          std::vector<hgx_block_results *> &results);

} // namespace hgx
