// What the column engine's library needs from the liftover engine (hal_amd/csrc/hgx_liftover.hip) to link in the host-side
// emulation (tests/cpp/hipshim): the liftover engine itself is NOT emulated — its kernels speak to the wavefront through DPP,
// readlane and ballots — so every entry into it says so.  Test infrastructure; never part of libhgx.so.
#include "hgx_liftover_engine.hpp"
#include <stdexcept>

namespace hgx {
static void no() {
    throw std::runtime_error("the liftover engine is not part of the host-side emulation (column engine only)");
}
hgx_liftover_plan *createLiftoverPlan(hgx_alignment *, int, int, const hgx_liftover_opts &, size_t, bool, bool) { no(); return nullptr; }
void submitLiftoverPlan(hgx_liftover_plan *, size_t, const int64_t *, const int64_t *, const uint8_t *, void *) { no(); }
void collectLiftoverPlan(hgx_liftover_plan *, const hgx_record **, size_t *) { no(); }
void runLiftoverPlan(hgx_liftover_plan *, size_t, const int64_t *, const int64_t *, const uint8_t *, void *, const hgx_record **, size_t *) { no(); }
void destroyLiftoverPlan(hgx_liftover_plan *) {}
const hgx_liftover_stats &liftoverPlanStats(const hgx_liftover_plan *) { no(); static hgx_liftover_stats s; return s; }
std::string liftoverPlanKernelTimes(hgx_liftover_plan *) { no(); return ""; }
std::string liftoverBuildPhases() { return "[]"; }
void liftoverPlanSetTiming(hgx_liftover_plan *, int) { no(); }
void liftoverPlanSetWorkers(hgx_liftover_plan *, int) { no(); }
void liftoverPlanCopyRecords(const hgx_liftover_plan *, void *, size_t, void *) { no(); }
void liftoverPlanCopyRecordsPacked(const hgx_liftover_plan *, void *, size_t, void *) { no(); }
size_t liftoverPlanWireBlob(hgx_liftover_plan *, void *, size_t, int64_t, int *, void *) { no(); return 0; }
void liftoverBatchHost(hgx_alignment *, int, int, size_t, const hgx_interval *, const hgx_liftover_opts &, std::vector<hgx_record> &, hgx_liftover_stats *) { no(); }
void liftoverBatchHostRaw(hgx_alignment *, int, int, size_t, const hgx_interval *, const hgx_liftover_opts &, const std::function<hgx_record *(size_t)> &) { no(); }
void liftoverStageQueries(hgx_alignment *, size_t, int64_t **, int64_t **, uint8_t **) { no(); }
void liftoverBatchStaged(hgx_alignment *, int, int, size_t, const hgx_liftover_opts &, const hgx_record **, size_t *, hgx_liftover_stats *, PackedRecords *) { no(); }
const ComposedUp *wholePathTable(hgx_alignment *, int, int, int) { no(); return nullptr; }
void liftoverBatchAbsolute(hgx_alignment *, int, int, const std::vector<int64_t> &, const std::vector<int64_t> &, const std::vector<uint8_t> &,
                           const hgx_liftover_opts &, std::vector<hgx_record> &) { no(); }
void blockMapHost(hgx_alignment *, int, int, int64_t, int64_t, bool, const hgx_liftover_opts &, std::vector<hgx_record> &) { no(); }
void ensureChainTables(const Image &, DeviceImage &, int, bool, bool) { no(); }
void ensureLocateTable(const Image &, DeviceImage &, int, int) { no(); }
void ensureDownTable(const Image &, DeviceImage &, int, int) { no(); }
} // namespace hgx

hgx_alignment::~hgx_alignment() {
    mafTracks.reset();
}
