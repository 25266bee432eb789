// Liftover engine: device image upload, the per-(source,target) walk schedule, workspaces, and the
// launch sequence behind hgx_liftover_run_device / hgx_liftover_batch (include/hgx.h).
#include "hgx_finish_kernel.hpp"
#include "hgx_liftover_engine.hpp"
#include "hgx_table_kernels.hpp"
#include "hgx_merged_kernels.hpp"
#include "hgx_lift_kernels.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <chrono>
#include <functional>
#include <cstdlib>
#include <mutex>
#include <cstring>
#include <map>

namespace hgx {

#define HIP_OK(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);               \
    } while (0)

template <typename C> static void *makeChainTable(const DeviceGenome &G, const DeviceGenome &P, bool last, size_t &bytes) {
    void *d = nullptr;
    const size_t n = (size_t)std::max<int64_t>(1, G.numTop);
    HIP_OK(hipMalloc(&d, n * sizeof(ChainRec<C>)));
    if (G.numTop > 0)
        hipLaunchKernelGGL((k_make_chain<C>), dim3((unsigned)((G.numTop + 255) / 256)), dim3(256), 0, nullptr, (const TopRec<C> *)G.top,
                           (const BotRec<C> *)P.bot, (uint32_t)G.numTop, last ? 1 : 0, (ChainRec<C> *)d);
    bytes += n * sizeof(ChainRec<C>);
    return d;
}

void ensureChainTables(const Image &img, DeviceImage &D, int genome, bool mid, bool last) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    const GenomeTables &G = img.genomes[(size_t)genome];
    if (G.parent < 0)
        throw std::runtime_error("ensureChainTables: genome has no parent");
    DeviceGenome &dg = D.genomes[(size_t)genome];
    const DeviceGenome &pg = D.genomes[(size_t)G.parent];
    HIP_OK(hipSetDevice(D.device));
    if (mid && !dg.chainMid)
        dg.chainMid = D.wide ? makeChainTable<int64_t>(dg, pg, false, D.bytes) : makeChainTable<int32_t>(dg, pg, false, D.bytes);
    if (last && !dg.chainLast)
        dg.chainLast = D.wide ? makeChainTable<int64_t>(dg, pg, true, D.bytes) : makeChainTable<int32_t>(dg, pg, true, D.bytes);
}

template <typename C> static void *makeDownTable(const DeviceGenome &P, const DeviceGenome &G, int slot, size_t &bytes) {
    void *d = nullptr;
    const size_t n = (size_t)std::max<int64_t>(1, P.numBot);
    HIP_OK(hipMalloc(&d, n * sizeof(DownRec<C>)));
    if (P.numBot > 0)
        hipLaunchKernelGGL((k_make_down<C>), dim3((unsigned)((P.numBot + 255) / 256)), dim3(256), 0, nullptr, (const int32_t *)P.childEnc[(size_t)slot],
                           (const TopRec<C> *)G.top, (uint32_t)P.numBot, (DownRec<C> *)d);
    bytes += n * sizeof(DownRec<C>);
    return d;
}

// coarse[b] = index of the segment holding position b << shift; ~4 segments per bucket, at most 2^22 buckets
void ensureLocateTable(const Image &img, DeviceImage &D, int genome, int which) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    DeviceGenome &dg = D.genomes[(size_t)genome];
    if (dg.locate[which])
        return;
    const GenomeTables &G = img.genomes[(size_t)genome];
    const int64_t nseg = which == 0 ? G.numTop : G.numBot;
    if (nseg <= 0 || G.totalLength <= 0)
        return;
    int64_t buckets = 1;
    while (buckets < nseg / 4 && buckets < ((int64_t)1 << 22))
        buckets <<= 1;
    int shift = 0;
    while (((G.totalLength - 1) >> shift) >= buckets)
        ++shift;
    const int64_t nb = ((G.totalLength - 1) >> shift) + 1;
    HIP_OK(hipSetDevice(D.device));
    HIP_OK(hipMalloc((void **)&dg.locate[which], ((size_t)nb + 1) * 4));
    const unsigned grid = (unsigned)((nb + 1 + 255) / 256);
    if (which == 0) {
        if (D.wide)
            hipLaunchKernelGGL((k_make_locate<TopRec<int64_t>>), dim3(grid), dim3(256), 0, nullptr, (const TopRec<int64_t> *)dg.top, nseg, shift, (uint32_t)nb, dg.locate[which]);
        else
            hipLaunchKernelGGL((k_make_locate<TopRec<int32_t>>), dim3(grid), dim3(256), 0, nullptr, (const TopRec<int32_t> *)dg.top, nseg, shift, (uint32_t)nb, dg.locate[which]);
    } else {
        if (D.wide)
            hipLaunchKernelGGL((k_make_locate<BotRec<int64_t>>), dim3(grid), dim3(256), 0, nullptr, (const BotRec<int64_t> *)dg.bot, nseg, shift, (uint32_t)nb, dg.locate[which]);
        else
            hipLaunchKernelGGL((k_make_locate<BotRec<int32_t>>), dim3(grid), dim3(256), 0, nullptr, (const BotRec<int32_t> *)dg.bot, nseg, shift, (uint32_t)nb, dg.locate[which]);
    }
    dg.locateShift[which] = shift;
    D.bytes += ((size_t)nb + 1) * 4;
}

void ensureDownTable(const Image &img, DeviceImage &D, int parent, int slot) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    const GenomeTables &P = img.genomes[(size_t)parent];
    DeviceGenome &dg = D.genomes[(size_t)parent];
    if (dg.downRec.size() < P.children.size())
        dg.downRec.resize(P.children.size(), nullptr);
    if (dg.downRec[(size_t)slot])
        return;
    HIP_OK(hipSetDevice(D.device));
    const DeviceGenome &cg = D.genomes[(size_t)P.children[(size_t)slot]];
    dg.downRec[(size_t)slot] = D.wide ? makeDownTable<int64_t>(dg, cg, slot, D.bytes) : makeDownTable<int32_t>(dg, cg, slot, D.bytes);
}


// ---------------------------------------------------------------------------------------------
// plan
struct DevBuf { // a workspace from the device's block cache (devAlloc / devRelease, hgx_device.hpp)
    void *p = nullptr;
    size_t n = 0;
    void ensure(size_t bytes) {
        if (bytes <= n)
            return;
        devRelease(p);
        p = nullptr;
        n = 0;
        p = devAlloc(bytes);
        n = bytes;
    }
    ~DevBuf() {
        devRelease(p);
    }
};

struct KernelTimer {
    struct Rec {
        std::string name;
        hipEvent_t a, b;
        int launch; // index of this launch's {top, bottom} deref counters, or -1
        unsigned long long top = 0, bot = 0;
    };
    struct Total {
        double ms = 0;
        int launches = 0;
        unsigned long long top = 0, bot = 0;
    };
    // mode 0: no events at all; 1: the kernel times of the last run (default); 2: accumulated over all runs since the last
    // read.  Elapsed times are only queried when somebody asks (hipEventElapsedTime per launch is host time inside a step).
    int mode = 1;
    std::vector<Rec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t used = 0, runStart = 0;
    std::map<std::string, Total> totals;
    void beginRun() {
        if (mode != 2) { // forget the previous run
            recs.clear();
            used = 0;
        }
        runStart = recs.size();
    }
    void begin(const char *name, hipStream_t s, int launch = -1) {
        if (mode == 0)
            return;
        if (used == pool.size()) {
            hipEvent_t a, b;
            HIP_OK(hipEventCreate(&a));
            HIP_OK(hipEventCreate(&b));
            pool.emplace_back(a, b);
        }
        recs.push_back({name, pool[used].first, pool[used].second, launch});
        ++used;
        HIP_OK(hipEventRecord(recs.back().a, s));
    }
    void end(hipStream_t s) {
        if (mode == 0)
            return;
        HIP_OK(hipEventRecord(recs.back().b, s));
    }
    void dropRun() { // the run is repeated (workspace regrown)
        recs.resize(runStart);
        used = runStart;
    }
    void endRun(const unsigned long long *hostCounters) { // keep this run's dereference counts with its launches
        for (size_t i = runStart; i < recs.size(); ++i)
            if (recs[i].launch >= 0) {
                recs[i].top = hostCounters[CNT_KSTAT0 + 2 * recs[i].launch];
                recs[i].bot = hostCounters[CNT_KSTAT0 + 2 * recs[i].launch + 1];
            }
    }
    const std::map<std::string, Total> &read() { // resolves what is pending; in mode 2 the window restarts
        totals.clear();
        for (Rec &r : recs) {
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, r.a, r.b));
            Total &t = totals[r.name];
            t.ms += ms;
            t.launches += 1;
            t.top += r.top;
            t.bot += r.bot;
        }
        if (mode == 2) {
            recs.clear();
            used = 0;
            runStart = 0;
        }
        return totals;
    }
    ~KernelTimer() {
        for (auto &p : pool) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    }
};

// Where the time of a plan's change-over to its table goes (HGX_BUILD_TIMING=1: printed to stderr, =2: kept for
// hgx_liftover_build_phases): the device is synchronised at every lap, so the phases add up to the wall time of the instrumented
// run (which is a little longer than an un-instrumented one).  One log per process; builds are serialised by ensureComposed's mutex.
struct BuildPhases {
    bool on = false, print = false;
    std::chrono::steady_clock::time_point last;
    std::vector<std::pair<std::string, double>> v;
    void start() {
        const char *e = getenv("HGX_BUILD_TIMING");
        on = e != nullptr;
        print = on && e[0] != '2';
        v.clear();
        if (on) {
            (void)hipDeviceSynchronize();
            last = std::chrono::steady_clock::now();
        }
    }
    void lap(const char *what) {
        if (!on)
            return;
        (void)hipDeviceSynchronize();
        const auto t = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t - last).count();
        if (print)
            fprintf(stderr, "[hgx build] %-40s %.3f ms\n", what, ms);
        v.emplace_back(what, ms);
        last = t;
    }
};
static BuildPhases g_phases;

std::string liftoverBuildPhases() {
    std::string s = "[";
    for (size_t i = 0; i < g_phases.v.size(); ++i) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s[\"%s\", %.4f]", i ? ", " : "", g_phases.v[i].first.c_str(), g_phases.v[i].second);
        s += buf;
    }
    return s + "]";
}

} // namespace hgx

using namespace hgx;

struct hgx_liftover_plan {
    hgx_alignment *h = nullptr; // not owned; must outlive every run (destroying the plan does not touch it)
    int device = 0;
    int src = -1, tgt = -1, mrca = -1;
    hgx_liftover_opts opts{};
    std::vector<int> up;                        // src ... mrca
    std::vector<std::pair<int, int>> down;      // (parent genome, child slot) per downward hop
    bool srcTop = true;
    const ComposedUp *composed = nullptr;       // composed up table src -> mrca (large plans; see ensureComposedUp), or null
    bool captureUp = false;                     // table builder: keep the pieces that arrive in the MRCA
    bool captureFinal = false;                  // table builder: keep the FINAL pieces (after the last down hop)
    bool composedThrough = false;               // which table this plan would use
    unsigned long long composedAfter = ~0ull;   // switch to the table once this many intervals have been walked (~0: never)
    unsigned long long walked = 0;
    int capturedBuf = -1, capturedLevel = -1;   // frontier and counter block of the captured pieces (the run stops there)
    std::vector<int> climb;                     // mrca ... coalescenceLimit when the limit lies above the MRCA (and dupes are on), else empty
    int numFrontiers = 2;
    bool levelSyncUp = getenv("HGX_LEVEL_SYNC_UP") != nullptr; // one launch per up level instead of k_up_chain (kept for deep trees and as a cross-check)
    size_t maxQueries = 0;
    uint32_t cap = 0; // piece capacity of every frontier / mapped / record buffer
    DevBuf fr[6][6], mp[2][6], counters, perQuery, offset, cursor, nOut, outOffset, blockSums, total, grouped, outRecords,
        deferredList, needCap, bigSlot, scratch, bigRecords, classLists, classCounts;
    // single-pass path over the merged table (hgx_lift_kernels.hpp)
    DevBuf liftKb, liftStatus, liftWorkMask, liftWorkCounts, liftExtraAlt;
    int liftLaunches = 0;        // launches of the last single-pass run that keep statistics
    bool liftWaveFinish = true;
    // hgx_liftover_submit / _collect: a batch whose launches are queued (1) or that has been run to the end already (2)
    int pendingState = 0;
    size_t pendingN = 0;
    const int64_t *pendingS = nullptr, *pendingE = nullptr;
    const uint8_t *pendingStrand = nullptr;
    hipStream_t pendingStream = nullptr;
    const hgx_record *pendingOut = nullptr;
    size_t pendingCount = 0;
    bool mergedOffThisRun = false;
    int liftGrid = 0, liftMinWaves = 1;
    unsigned long long generalQueries = 0, liftRestCount = 0;
    bool liftRestSeen = false, liftRestSkipped = false; // (see runMergedOnce: the launches behind k_lift_classify for what it passes on)
    bool liftStateClean = false; // the counters are zero (left so by the last run's k_lift_totals)
    size_t liftTilesCap = 0, liftGroupsCap = 0; // tiles, groups of 64 tiles liftStatus has room for
    uint32_t liftWorkers = 0;    // workgroups of k_lift_classify that go for the general intervals first (from the last run's count)
    unsigned long long liftLastQueries = 0;
    int liftWorkersWanted = -1;  // hgx_liftover_plan_set_workers
    // scouts (k_lift_classify): two sets of the words the general intervals' lines are counted in — {liftStatus's, liftExtraAlt} —, a
    // batch counts in one and its tiles clear the other for the batch behind it.  liftExtraDirty[k]: the tiles set k may hold
    // counts for (0: zero throughout); liftExtraSet: the set the next scout batch counts in
    size_t liftExtraDirty[2] = {0, 0};
    int liftExtraSet = 0;
    // scratch of the single-pass runs for intervals that outgrow the LDS finishing kernel (k_finish_big without a host
    // synchronisation in between): liftBigSlots slices for liftBigCap pieces each, grown when a run needed more
    uint32_t liftBigSlots = 0;
    int liftBigCap = 0;
    KernelTimer timer;
    DevBuf wireFlag;                      // hgx_liftover_wire_blob: "a field does not fit the 12-byte form"
    unsigned int *wireFlagHost = nullptr;
    unsigned long long *pinned = nullptr; // host-pinned copy of the counter block + the record total (one readback per run)
    unsigned long long *pinnedDev = nullptr; // the same memory as the device sees it (k_lift_totals writes its report there)
    hgx_liftover_stats stats{};
    hipEvent_t evStart = nullptr, evWalk = nullptr, evEnd = nullptr;
    ~hgx_liftover_plan() {
        if (pinned)
            (void)hipHostFree(pinned);
        if (wireFlagHost)
            (void)hipHostFree(wireFlagHost);
        if (evStart)
            (void)hipEventDestroy(evStart);
        if (evWalk)
            (void)hipEventDestroy(evWalk);
        if (evEnd)
            (void)hipEventDestroy(evEnd);
    }
    Frontier frontier(int k) {
        return Frontier{(int32_t *)fr[k][0].p, (int64_t *)fr[k][1].p, (int32_t *)fr[k][2].p,
                        (int64_t *)fr[k][3].p, (int32_t *)fr[k][4].p, (uint8_t *)fr[k][5].p};
    }
    Mapped mapped(int k) {
        return Mapped{(MappedRec *)mp[k][0].p};
    }
    // The frontiers and the first mapped-piece buffer are only needed by runs that walk (level by level, the up table, the
    // table builder); a plan served from a whole-path table never touches them, and they are most of a plan's memory —
    // allocating gigabytes is what a fresh plan's first batch would otherwise wait for.
    bool walkBuffers = false;
    bool tableBuilder = false; // created by buildComposed: walks every source segment up to the capture point
    void ensureWalkBuffers() {
        if (walkBuffers)
            return;
        walkBuffers = true;
        allocate(cap);
    }
    void allocate(uint32_t newCap) {
        cap = newCap;
        liftStateClean = false; // (buffers may move: their contents are not what the last run's epilogue left)
        static const size_t fsz[6] = {4, 8, 4, 8, 4, 1};
        if (walkBuffers) {
            for (int k = 0; k < numFrontiers; ++k)
                for (int a = 0; a < 6; ++a)
                    fr[k][a].ensure(fsz[a] * (size_t)cap);
            mp[0][0].ensure(sizeof(MappedRec) * (size_t)cap);
        }
        if (!tableBuilder) { // (a table builder's run ends at the captured frontier: no pieces grouped by interval, no records)
            mp[1][0].ensure(sizeof(MappedRec) * (size_t)cap);
            grouped.ensure(sizeof(hgx_record) * ((size_t)cap + (size_t)liftBigSlots * (size_t)liftBigCap));
            outRecords.ensure(sizeof(hgx_record) * (size_t)cap);
        }
        if (liftBigSlots)
            scratch.ensure((h->dev->wide ? finishSliceBytes<int64_t>(liftBigCap) : finishSliceBytes<int32_t>(liftBigCap)) * (size_t)liftBigSlots);
        if (!pinned) {
            HIP_OK(hipHostMalloc((void **)&pinned, 8 * (CNT_SLOTS + 1 + LIFT_RB_WORDS), hipHostMallocMapped));
            HIP_OK(hipHostGetDevicePointer((void **)&pinnedDev, pinned, 0));
        }
        const size_t nq = std::max<size_t>(maxQueries, 1);
        counters.ensure(8 * (CNT_DEV_SLOTS + LIFT_RB_WORDS));
        perQuery.ensure(4 * (nq + 1));
        offset.ensure(4 * (nq + 1));
        cursor.ensure(4 * (nq + 1));
        nOut.ensure(4 * (nq + 1));
        outOffset.ensure(4 * (nq + 1));
        blockSums.ensure(4 * (nq / SCAN_BLOCK + 2));
        total.ensure(16);
        deferredList.ensure(4 * (nq + 1));
        needCap.ensure(4 * (nq + 1));
        bigSlot.ensure(4 * (nq + 1));
        classLists.ensure(4 * 4 * (nq + 1));
        classCounts.ensure(32);
        liftKb.ensure(8 * (nq + 1));
        liftTilesCap = std::max(liftTilesCap, (nq + LIFT_TILE - 1) / LIFT_TILE);
        // [lines k_lift_classify's workers add per 64 intervals | the same per group of 64 tiles | lines per 64 intervals | lines per group]
        liftGroupsCap = std::max(liftGroupsCap, (nq + 64 * LIFT_TILE - 1) / (64 * LIFT_TILE));
        liftStatus.ensure(32 * liftTilesCap + 16 * liftGroupsCap + 16);
        liftExtraAlt.ensure(16 * liftTilesCap + 8 * liftGroupsCap + 16);
        liftWorkMask.ensure(8 * ((nq + 63) / 64 + 1));
        liftWorkCounts.ensure(8 * LIFT_LISTS * LIFT_LIST_PITCH);
    }
};

namespace hgx {

static constexpr int GRID = 2048; // 256 CUs x 8 resident 256-thread blocks, grid-stride beyond

// Grid of a grid-stride kernel that does not fit eight blocks per CU (more than 64 VGPRs): exactly the resident blocks.
// With GRID blocks the ones that do not fit start when the first ones finish and run their whole share on a nearly idle
// chip — k_locate_through (68 VGPRs, 7 blocks per CU) took twice the lifetime of its wavefronts (PMC, profiles/r01q_pmc.txt).
template <typename K> static int residentGrid(K kernel, size_t dynLds = 0) {
    static std::mutex mu;
    static std::map<std::pair<const void *, size_t>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair((const void *)kernel, dynLds);
    auto it = cache.find(key);
    if (it != cache.end())
        return it->second;
    int perCu = 0, dev = 0;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDevice(&dev));
    HIP_OK(hipGetDeviceProperties(&prop, dev));
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, 256, dynLds));
    const int g = std::min(GRID, std::max(1, perCu) * prop.multiProcessorCount);
    cache.emplace(key, g);
    return g;
}

// may k_finish_big stage an interval's slice in LDS?  Up to the 160 KB a workgroup can have (beyond 64 KB the kernel is told once)
template <typename C> static bool bigLdsOk(size_t sliceBytes) {
    if (getenv("HGX_BIG_IN_LDS") && getenv("HGX_BIG_IN_LDS")[0] == '0')
        return false;
    if (sliceBytes > 160 * 1024 - 1024)
        return false;
    static std::mutex mu;
    static size_t allowed = 64 * 1024;
    std::lock_guard<std::mutex> lock(mu);
    if (sliceBytes > allowed) {
        if (hipFuncSetAttribute((const void *)k_finish_big<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 1024)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        allowed = 160 * 1024 - 1024;
    }
    return true;
}

static void exclusiveScan(hgx_liftover_plan &P, const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *total, hipStream_t s) {
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    P.timer.begin("scan", s);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, s, in, n, (uint32_t *)P.blockSums.p);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, (uint32_t *)P.blockSums.p, nb, total);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, s, in, n, (const uint32_t *)P.blockSums.p, out);
    P.timer.end(s);
}

static const ComposedUp *ensureComposed(hgx_alignment *h, int src, int dst, bool through, const hgx_liftover_opts &opts, bool wantMerged = true);

// One batch on the single-pass path: classify, the general intervals through the unmerged table and the LDS finishing
// kernel, then k_lift_merged writes every record at its final place.  One host synchronisation at the end.  C: the coordinate
// type of the alignment's tables (int64 for alignments with a genome of 2^31 bases or more — hal_index_t, api/inc/halDefs.h:34).
static void finishMergedOnce(hgx_liftover_plan &P, hipStream_t s, unsigned long long *hostCounters);
// launchOnly: the launches are queued and the function returns; finishMergedOnce waits for them and reads the report
template <typename C>
static void runMergedOnce(hgx_liftover_plan &P, size_t n, const int64_t *dS, const int64_t *dE, const uint8_t *dStrand, hipStream_t s,
                          unsigned long long *hostCounters, bool launchOnly = false) {
    const DeviceImage &D = *P.h->dev;
    const ComposedUp &T = *P.composed;
    unsigned long long *cnt = (unsigned long long *)P.counters.p;
    const uint32_t cap = P.cap, nq = (uint32_t)n;
    P.liftLastQueries = n;
    const int64_t srcLength = P.h->img.genomes[(size_t)P.src].totalLength;
    const DeviceGenome &TG = D.genomes[(size_t)P.tgt];
    const uint32_t nTiles = (nq + LIFT_TILE - 1) / LIFT_TILE, nGroups = (nTiles + 63) / 64;
    // lines per 64 intervals (k_lift_classify, + the finishing kernels) and per group of 64 tiles (k_lift_totals)
    // (+ what k_lift_classify's workers count beside them, in a run that has workers)
    uint32_t *waveExtra = (uint32_t *)P.liftStatus.p;
    unsigned long long *groupExtra = (unsigned long long *)(waveExtra + 4 * P.liftTilesCap);
    uint32_t *waveTotal = (uint32_t *)(groupExtra + P.liftGroupsCap);
    unsigned long long *groupTotal = (unsigned long long *)waveTotal + 2 * (size_t)nTiles; // (behind the 4 * nTiles 32-bit words)
    uint32_t *generalList = (uint32_t *)P.classLists.p;
    unsigned long long *generalCount = (unsigned long long *)P.classCounts.p;
    // the words this run counts in were left zeroed by the previous single-pass run's epilogue; otherwise (first run, another
    // path in between, a repeated run) they are cleared here
    const bool wasClean = P.liftStateClean;
    P.liftStateClean = false;
    if (!wasClean) {
        HIP_OK(hipMemsetAsync(cnt, 0, 8 * CNT_DEV_SLOTS, s));
        HIP_OK(hipMemsetAsync(P.liftStatus.p, 0, P.liftStatus.n, s));
        HIP_OK(hipMemsetAsync(generalCount, 0, 32, s));
        HIP_OK(hipMemsetAsync(P.liftWorkCounts.p, 0, P.liftWorkCounts.n, s));
        HIP_OK(hipMemsetAsync(P.liftExtraAlt.p, 0, P.liftExtraAlt.n, s));
        P.liftExtraDirty[0] = P.liftExtraDirty[1] = 0;
    }
    const bool events = P.timer.mode != 0; // (walk_ms / total_ms of the statistics need three event records per run)
    if (events)
        HIP_OK(hipEventRecord(P.evStart, s));
    const GeneralTable<C> GT{(const uint32_t *)T.coarse, (const uint32_t *)T.starts, T.shift, (const ComposedRec<C> *)T.recs, srcLength,
                             (const int64_t *)TG.seqStart, (int)TG.numSeq,
                             P.h->img.genomes[(size_t)P.tgt].seqs.empty() ? 0 : (int64_t)P.h->img.genomes[(size_t)P.tgt].seqs[0].start,
                             (hgx_record *)P.grouped.p, cap, cnt + CNT_FRONT0, cnt};
    uint32_t *restList = generalList + nq;
    unsigned long long *restCount = generalCount + 1;
    const bool waveFinish = !(getenv("HGX_FINISH_WAVE") && getenv("HGX_FINISH_WAVE")[0] == '0');
    int launch = 0;
    auto kstat = [&]() { return cnt + CNT_DSTAT0 + STAT_LAUNCH0 + 2 * launch; };
    // k_lift_classify: every interval's reach and its number of lines; the general intervals are finished by the wavefronts that
    // meet them (HGX_FINISH_WAVE=0: they are all listed instead and go the LDS way, the round-1 route, as a cross-check)
    const uint32_t *lateList = waveFinish ? restList : generalList;
    unsigned long long *lateCount = waveFinish ? restCount : generalCount;
    // (the launches behind k_lift_classify for what it passes on are made only once a run of this plan has passed something on)
    P.liftRestSkipped = waveFinish && !P.liftRestSeen;
    const int storeLaunch = P.liftRestSkipped ? 1 : 2; // k_lift_merged's place among the launches that keep statistics
    // workers: the last run's general intervals were few enough to be looked for up front (HGX_LIFT_WORKERS: their number, 0 = none)
    // (a batch that is launched and left — others are in flight beside it — overlaps its launches' tails with theirs: the pass
    // over its intervals would be work added, not time saved)
    uint32_t workers = !waveFinish ? 0u : P.liftWorkersWanted >= 0 ? (uint32_t)P.liftWorkersWanted : launchOnly ? 0u : P.liftWorkers;
    if (const char *e = getenv("HGX_LIFT_WORKERS"))
        workers = waveFinish ? (uint32_t)std::max(0, atoi(e)) : 0u;
    // (the lists: the third and fourth quarter of classLists, a list's room = what its workgroups look at and more)
    uint32_t *workList = generalList + 2 * (size_t)nq;
    const uint32_t workListCap = (((nq + 511) / 512 + LIFT_LISTS - 1) / LIFT_LISTS) * 512;
    unsigned long long *workCounts = (unsigned long long *)P.liftWorkCounts.p;
    if (workers && (size_t)workListCap * LIFT_LISTS > 2 * (size_t)nq + 2)
        workers = 0; // (a batch of a few hundred intervals: no room for 64 lists, and nothing to gain)
    // scouts (round 6): the workers find their intervals themselves (HGX_LIFT_SCOUT=0: k_lift_general_list's pass in front, round
    // 3's form).  A share is a round of LIFT_SCOUT_ROUND intervals where the batch allows, so that a scout's intervals are all
    // under way behind one look.  Measured at config 2 (profiles/r06w_scout_diag.txt), a batch by itself: 0.1068 ms against 0.1078
    // with the pass and 0.1153 inline (k_lift_classify 0.0435 / 0.0365 + 0.0104 / 0.0527).  Batches launched and left still go
    // inline: with two in flight the step is the sum of the launches' bodies (0.0846 ms), and the scouts' look at the batch is work
    // added to it (0.0889)
    const bool scoutsAllowed = !(getenv("HGX_LIFT_SCOUT") && getenv("HGX_LIFT_SCOUT")[0] == '0');
    uint32_t scoutShare = 0;
    uint32_t *otherWaveExtra = nullptr;
    unsigned long long *otherGroupExtra = nullptr;
    if (workers && scoutsAllowed) {
        uint32_t most = 4096u; // (HGX_LIFT_SCOUT_MAX: fewer, so that a scout's share takes several rounds — the tests' way to those)
        if (const char *e = getenv("HGX_LIFT_SCOUT_MAX"))
            most = (uint32_t)std::max(1, atoi(e));
        workers = std::min<uint32_t>(most, std::max<uint32_t>(workers, (nq + LIFT_SCOUT_ROUND - 1) / LIFT_SCOUT_ROUND));
        scoutShare = (((nq + workers - 1) / workers) + 255u) & ~255u;
        workers = (nq + scoutShare - 1) / scoutShare;
        const int k = P.liftExtraSet;
        uint32_t *altWave = (uint32_t *)P.liftExtraAlt.p;
        unsigned long long *altGroup = (unsigned long long *)(altWave + 4 * P.liftTilesCap);
        if (P.liftExtraDirty[k]) { // (the batch that should have cleared it was a shorter one, or not a scout batch)
            if (k == 0) {
                HIP_OK(hipMemsetAsync(waveExtra, 0, 16 * P.liftTilesCap + 8 * P.liftGroupsCap, s));
            } else {
                HIP_OK(hipMemsetAsync(altWave, 0, 16 * P.liftTilesCap + 8 * P.liftGroupsCap, s));
            }
        }
        otherWaveExtra = k == 0 ? altWave : waveExtra;
        otherGroupExtra = k == 0 ? altGroup : groupExtra;
        if (k == 1) {
            waveExtra = altWave;
            groupExtra = altGroup;
        }
        P.liftExtraDirty[k] = nTiles;
        if (P.liftExtraDirty[1 - k] <= nTiles)
            P.liftExtraDirty[1 - k] = 0;
        P.liftExtraSet = 1 - k;
    } else if (workers) {
        P.liftExtraDirty[0] = std::max(P.liftExtraDirty[0], (size_t)nTiles); // (k_lift_general_list clears what this batch counts in, and no more)
    }
    if (workers && !scoutShare) {
        P.timer.begin("k_lift_general_list", s);
        hipLaunchKernelGGL(k_lift_general_list, dim3((nq + 511) / 512), dim3(256), 0, s, dS, dE, nq, srcLength, (const uint32_t *)T.mFlagBits, T.mShift,
                           T.mWindow, (unsigned long long *)P.liftWorkMask.p, workList, workListCap, workCounts, waveExtra, groupExtra);
        P.timer.end(s);
    }
    P.timer.begin("k_lift_classify", s, launch);
#define HGX_CLASSIFY(INL, W)                                                                                                                 \
    hipLaunchKernelGGL((k_lift_classify<C, INL, W>), dim3(std::max<uint32_t>(1, nTiles) + workers), dim3(256), 0, s, dS, dE, dStrand, nq,     \
                       srcLength, (const uint32_t *)T.mBuckets, T.mShift, T.mWindow, (const ComposedRec<C> *)T.mRecs, (uint2 *)P.liftKb.p, GT,\
                       kstat(), cnt + CNT_DSTAT0 + STAT_LAUNCH0 + 2 * storeLaunch, (uint32_t *)P.offset.p, (uint32_t *)P.nOut.p,              \
                       (uint32_t *)lateList, lateCount, waveTotal, workers, waveExtra, (const unsigned long long *)P.liftWorkMask.p,         \
                       (const uint32_t *)workList, workListCap, (const unsigned long long *)workCounts, groupExtra, scoutShare,                \
                       (const uint32_t *)T.mFlagBits, (uint32_t)(T.mNum + LIFT_SENTINELS - 1), otherWaveExtra, otherGroupExtra)
    if (!waveFinish) {
        HGX_CLASSIFY(false, 1);
    } else if constexpr (sizeof(C) == 8) {
        HGX_CLASSIFY(true, 4); // (64-bit coordinates: twice the registers per record and per piece)
    } else {
        HGX_CLASSIFY(true, 6); // (80 VGPRs; compiled for 1 / 7 / 8 wavefronts per SIMD it took 0.056 / 0.058 / 0.085 ms against 0.054)
    }
#undef HGX_CLASSIFY
    P.timer.end(s);
    ++launch;
    // Two more launches for what k_lift_classify passes on (general intervals of more than 64 pieces) — made only once a run of this plan has passed
    // something on: a run that skips them reads the count back with its counters and is repeated with them when it is not zero
    // (runPlan), so batches without such intervals do not pay for two empty launches.
    if (!P.liftRestSkipped) {
    P.timer.begin("k_locate_through", s, launch);
    // (the list's length is only known on the device; the grids are sized for a list that is a small part of the batch)
    hipLaunchKernelGGL((k_locate_through<C>), dim3(std::min(512, residentGrid(k_locate_through<C>))), dim3(256), 0, s, dS, dE, dStrand, nq, srcLength,
                       (const uint32_t *)T.coarse, (const uint32_t *)T.starts, T.shift, (const ComposedRec<C> *)T.recs, P.mapped(1), cap,
                       cnt + CNT_FRONT0, cnt, kstat(), (uint32_t *)P.offset.p, (uint32_t *)P.perQuery.p, lateList, lateCount);
    P.timer.end(s);
    ++launch;
    if (events)
        HIP_OK(hipEventRecord(P.evWalk, s));
    P.timer.begin("k_finish_lds", s);
    // (128 pieces of staging — 10 KB of LDS: fifteen intervals a CU instead of seven; a wavefront alone on its SIMD leaves most of its
    // issue slots empty, and the few intervals that outgrow 128 go on to k_finish_big, which stages in LDS as well)
    hipLaunchKernelGGL((k_finish_lds<C, 128>), dim3(std::min<uint32_t>(std::max<uint32_t>(nq, 1), 3840u)), dim3(64), 0, s, P.mapped(1),
                       (const uint32_t *)P.offset.p, (const uint32_t *)P.perQuery.p, lateList, lateCount, (const int64_t *)TG.seqStart,
                       (int)TG.numSeq, (hgx_record *)P.grouped.p, (uint32_t *)P.nOut.p, (uint32_t *)P.deferredList.p, (uint32_t *)P.needCap.p,
                       cnt, 0, waveTotal, cnt + CNT_DSTAT0 + STAT_LAUNCH0 + 2 * storeLaunch + 1);
    P.timer.end(s);
    } else if (events) {
        HIP_OK(hipEventRecord(P.evWalk, s));
    }
    if (!P.liftRestSkipped && getenv("HGX_LATE_HISTOGRAM")) { // diagnostics: how many pieces the intervals passed on have (stderr)
        HIP_OK(hipStreamSynchronize(s));
        unsigned long long nLate = 0;
        HIP_OK(hipMemcpy(&nLate, lateCount, 8, hipMemcpyDeviceToHost));
        std::vector<uint32_t> list((size_t)nLate), per((size_t)nq + 1);
        if (nLate)
            HIP_OK(hipMemcpy(list.data(), lateList, 4 * (size_t)nLate, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(per.data(), P.perQuery.p, 4 * ((size_t)nq + 1), hipMemcpyDeviceToHost));
        unsigned long long hist[16] = {0}, pieces = 0, mx = 0;
        for (uint32_t q : list) {
            const uint32_t c = per[q];
            int b = 0;
            while ((1u << b) < c && b < 15)
                ++b;
            ++hist[b];
            pieces += c;
            mx = std::max<unsigned long long>(mx, c);
        }
        fprintf(stderr, "[hgx late] %llu intervals passed on, %llu pieces, largest %llu; by pieces <= 2^b:", nLate, pieces, mx);
        for (int b = 0; b < 16; ++b)
            fprintf(stderr, " %d:%llu", b, hist[b]);
        fprintf(stderr, "\n");
    }
#if defined(HGX_LIFT_PROFILE) && HGX_LIFT_PROFILE == 4
    if (!P.liftRestSkipped) {
        HIP_OK(hipStreamSynchronize(s));
        unsigned long long fp[16];
        HIP_OK(hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_finishProfile), sizeof fp));
        fprintf(stderr, "(k_finish_lds) [hgx finish profile] Mcycles: sort %.2f | boundary sort %.2f | unique+prefix %.2f | emit %.2f | sort2 %.2f | dups+seq %.2f | extract %.2f | line sort %.2f | load %.2f\n",
                fp[0] / 1e6, fp[1] / 1e6, fp[2] / 1e6, fp[3] / 1e6, fp[4] / 1e6, fp[5] / 1e6, fp[6] / 1e6, fp[7] / 1e6, fp[8] / 1e6);
        memset(fp, 0, sizeof fp);
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_finishProfile), fp, sizeof fp));
    }
#endif
    if (P.liftBigSlots && !P.liftRestSkipped) { // intervals k_finish_lds deferred: same algorithm on global scratch; their records become slices behind the grouped buffer
        P.timer.begin("k_finish_big", s);
        const size_t bigSlice = finishSliceBytes<C>(P.liftBigCap);
        const int bigInLds = bigLdsOk<C>(bigSlice) ? 1 : 0; // (a few hundred pieces: staged in LDS, not in the global scratch)
        hipLaunchKernelGGL((k_finish_big<C>), dim3(P.liftBigSlots), dim3(64), bigInLds ? bigSlice : 0, s, P.mapped(1), (const uint32_t *)P.offset.p,
                           (const uint32_t *)P.perQuery.p, (const uint32_t *)P.deferredList.p, P.liftBigSlots, P.liftBigCap,
                           (unsigned char *)P.scratch.p, bigSlice, (const int64_t *)TG.seqStart, (int)TG.numSeq,
                           (hgx_record *)P.grouped.p + cap, (uint32_t *)P.nOut.p, cnt, 0, 1, (uint32_t *)P.offset.p, cap, waveTotal,
                           cnt + CNT_DSTAT0 + STAT_LAUNCH0 + 2 * storeLaunch + 1, bigInLds);
        P.timer.end(s);
    }
    // everything else, and the dense output
    if (!P.liftGrid) {
        const char *v = getenv("HGX_LIFT_MINWAVES"); // 8: hold the kernel to 64 VGPRs (experiments)
        P.liftMinWaves = v && atoi(v) == 8 ? 8 : 1;
        P.liftGrid = 1 << 20; // a workgroup per tile, the hardware deals them out (HGX_LIFT_BLOCKS: a looping grid of that many per CU)
        if (const char *b = getenv("HGX_LIFT_BLOCKS"))
            P.liftGrid = std::max(1, atoi(b)) * 256;
    }
    // all lines, the groups' lines, the statistics — and everything the host wants to know goes to pinned memory from here
    P.timer.begin("k_lift_totals", s);
    hipLaunchKernelGGL(k_lift_totals, dim3(1), dim3(1024), 0, s, (const uint32_t *)waveTotal, workers ? (const unsigned long long *)groupExtra : nullptr, 4 * nTiles, nGroups,
                       groupTotal, cap, (uint32_t *)P.total.p, cnt, generalCount, waveFinish ? restCount : (unsigned long long *)nullptr, workCounts,
                       storeLaunch, P.pinnedDev + CNT_SLOTS + 1);
    P.timer.end(s);
    // a round's records staged in LDS and stored as the one span of the output they are (lift_wave_emit) — HGX_LIFT_STAGED=0: every
    // record stored by its lane, 40 bytes 40 bytes apart (before round 6's last day; config 4's launch 0.50 -> 0.377 ms, its step 0.84 ->
    // 0.71 ms; config 2's 0.053 -> 0.049 ms: profiles/r06ac_staged_stores.txt)
    const int stagedStores = getenv("HGX_LIFT_STAGED") && getenv("HGX_LIFT_STAGED")[0] == '0' ? 0 : 1;
    P.timer.begin("k_lift_merged", s, launch);
#define HGX_LIFT(W)                                                                                                                          \
    hipLaunchKernelGGL((k_lift_merged<C, W>), dim3(std::min<uint32_t>((uint32_t)P.liftGrid, nTiles)), dim3(256), 0, s, dS, dE, dStrand, nq,     \
                       (const uint2 *)P.liftKb.p, (const ComposedRec<C> *)T.mRecs, (const int64_t *)TG.seqStart, (int)TG.numSeq,             \
                       (const uint32_t *)P.offset.p, (const hgx_record *)P.grouped.p, (hgx_record *)P.outRecords.p, cap,                     \
                       (const uint32_t *)P.nOut.p, (uint32_t *)P.outOffset.p, waveTotal, workers ? (const uint32_t *)waveExtra : nullptr, groupTotal, \
                       nTiles, stagedStores)
    if (P.liftMinWaves == 8)
        HGX_LIFT(8);
    else
        HGX_LIFT(1);
#undef HGX_LIFT
    P.timer.end(s);
    ++launch;
#if defined(HGX_LIFT_PROFILE) && HGX_LIFT_PROFILE == 4
    if (!P.liftRestSkipped) {
        fprintf(stderr, "(k_finish_big) ");
        HIP_OK(hipStreamSynchronize(s));
        unsigned long long fp[16];
        HIP_OK(hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_finishProfile), sizeof fp));
        fprintf(stderr, "[hgx finish profile] Mcycles: load+sort %.2f | boundary sort %.2f | unique+prefix %.2f | emit %.2f | sort2 %.2f | dups+seq %.2f | extract %.2f | line sort %.2f | load %.2f\n",
                fp[0] / 1e6, fp[1] / 1e6, fp[2] / 1e6, fp[3] / 1e6, fp[4] / 1e6, fp[5] / 1e6, fp[6] / 1e6, fp[7] / 1e6, fp[8] / 1e6);
        memset(fp, 0, sizeof fp);
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_finishProfile), fp, sizeof fp));
    }
#endif
    if (events)
        HIP_OK(hipEventRecord(P.evEnd, s));
    P.liftLaunches = launch;
    P.liftWaveFinish = waveFinish;
    if (!launchOnly)
        finishMergedOnce(P, s, hostCounters);
}

static void finishMergedOnce(hgx_liftover_plan &P, hipStream_t s, unsigned long long *hostCounters) {
    unsigned long long *cnt = (unsigned long long *)P.counters.p;
    unsigned long long *hrb = P.pinned + CNT_SLOTS + 1;
    HIP_OK(hipStreamSynchronize(s));
    memset(hostCounters, 0, 8 * CNT_SLOTS);
    if (hrb[CNT_OVERFLOW]) { // the retry sizes the buffers from the frontier counters and CNT_LIFT_TOTAL
        HIP_OK(hipMemcpyAsync(P.pinned, cnt, 8 * CNT_SLOTS, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        memcpy(hostCounters, P.pinned, 8 * CNT_SLOTS);
    }
    memcpy(hostCounters, hrb, 8 * 8);
    for (int l = 0; l < P.liftLaunches; ++l)
        hostCounters[CNT_KSTAT0 + 2 * l] = hrb[8 + l];
    P.pinned[CNT_SLOTS] = hrb[CNT_LIFT_TOTAL]; // the record total, where runPlan looks for it
    // (an overflowing or failing run may have left counters beyond the ones the epilogue clears)
    P.liftStateClean = !hrb[CNT_OVERFLOW] && !hrb[CNT_DEFERRED] && !(getenv("HGX_LIFT_MEMSETS") != nullptr);
    P.generalQueries = P.liftWaveFinish ? hrb[14] : hrb[12];
    // the next run's workers: about a wavefront per general interval of this one, when they are a small part of the batch
    // (a batch of long intervals is all general: the tiles' own wavefronts are the many hands then)
    {
        const unsigned long long g = P.generalQueries, nqLast = P.liftLastQueries;
        P.liftWorkers = g > 0 && g * 64 <= nqLast ? (uint32_t)std::min<unsigned long long>(1024, std::max<unsigned long long>(32, (g * 3 / 2 + 3) / 4)) : 0u;
    }
    P.liftRestCount = hrb[13];
}

template <typename C>
static void runOnce(hgx_liftover_plan &P, size_t n, const int64_t *dS, const int64_t *dE, const uint8_t *dStrand, hipStream_t s,
                    unsigned long long *hostCounters) {
    const DeviceImage &D = *P.h->dev;
    unsigned long long *cnt = (unsigned long long *)P.counters.p;
    const uint32_t cap = P.cap;
    const uint32_t nq = (uint32_t)n;
    if (P.composed && P.composed->mRecs && !P.opts.emit_blocks && !P.mergedOffThisRun && !P.captureUp && !P.captureFinal) {
        runMergedOnce<C>(P, n, dS, dE, dStrand, s, hostCounters);
        return;
    }
    P.liftStateClean = false; // (this run counts in words a single-pass run expects zeroed)
    HIP_OK(hipMemsetAsync(cnt, 0, 8 * CNT_DEV_SLOTS, s));
    if (!(P.composed && P.composed->through)) { // (k_locate_through writes every interval's count itself and has no grouping scatter)
        HIP_OK(hipMemsetAsync(P.perQuery.p, 0, 4 * (n + 1), s));
        HIP_OK(hipMemsetAsync(P.cursor.p, 0, 4 * (n + 1), s));
    }
    HIP_OK(hipEventRecord(P.evStart, s));

    if (!(P.composed && P.composed->through))
        P.ensureWalkBuffers();
    int level = 0;
    int launch = 0; // per-launch deref counter slot
    auto kstat = [&]() { return cnt + CNT_DSTAT0 + STAT_LAUNCH0 + 2 * launch; }; // (copy 0; a block adds to its own copy)
    int cur = 0; // frontier buffer holding the current pieces
    auto inCnt = [&]() { return cnt + CNT_FRONT0 + (size_t)level; };
    int outLevel = 1; // counter block of the frontier the next launch writes; every launch gets a fresh one
    auto outCnt = [&]() { return cnt + CNT_FRONT0 + (size_t)outLevel; };
    const int64_t minLen = P.opts.min_length;

    // stage 0
    const DeviceGenome &SG = D.genomes[(size_t)P.src];
    // (processing the batch sorted by start position was measured: -14 % on the walk kernels, but +0.17 ms in the per-interval
    // atomics of the grouping step and 0.19 ms for the sort itself — no net gain, so batches run in arrival order)
    const bool useComposed = P.composed != nullptr;
    const bool through = useComposed && P.composed->through; // the table's pieces are final: straight to the grouping step
    if (useComposed) {
        // locate + the whole up phase from the composed table: the pieces arrive in the MRCA directly
        P.timer.begin(through ? "k_locate_through" : "k_locate_composed", s, launch);
        if (through)
            hipLaunchKernelGGL((k_locate_through<C>), dim3(residentGrid(k_locate_through<C>)), dim3(256), 0, s, dS, dE, dStrand, nq,
                               P.h->img.genomes[(size_t)P.src].totalLength, (const uint32_t *)P.composed->coarse,
                               (const uint32_t *)P.composed->starts, P.composed->shift, (const ComposedRec<C> *)P.composed->recs, P.mapped(1), cap,
                               inCnt(), cnt, kstat(), (uint32_t *)P.offset.p, (uint32_t *)P.perQuery.p);
        else
            hipLaunchKernelGGL((k_locate_composed<C>), dim3(GRID), dim3(256), 0, s, dS, dE, dStrand, nq,
                               P.h->img.genomes[(size_t)P.src].totalLength, (const uint32_t *)P.composed->coarse, P.composed->shift,
                               (const ComposedRec<C> *)P.composed->recs, (const C *)P.composed->eo, (uint64_t)P.composed->numRecs, P.frontier(cur),
                               cap, inCnt(), cnt, kstat());
        P.timer.end(s);
        ++launch;
    } else {
        P.timer.begin("k_locate_expand", s, launch);
        if (P.srcTop)
            hipLaunchKernelGGL((k_locate_expand<TopRec<C>>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)SG.top, SG.numTop, dS, dE,
                               dStrand, nq, (const int32_t *)SG.locate[0], SG.locateShift[0], P.frontier(cur), cap, cnt, kstat() + 0);
        else
            hipLaunchKernelGGL((k_locate_expand<BotRec<C>>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)SG.bot, SG.numBot, dS, dE,
                               dStrand, nq, (const int32_t *)SG.locate[1], SG.locateShift[1], P.frontier(cur), cap, cnt, kstat() + 1);
        P.timer.end(s);
        ++launch;
    }

    bool curTop = P.srcTop;
    int curGenome = P.src;
    if (useComposed) {
        curGenome = P.mrca;
        curTop = false;
    } else if (P.src != P.mrca) {
        // up phase: P.up = [src, ..., mrca].  k_up_first lifts the source's top pieces to its parent; every further level
        // is one k_up_walk launch.  The last launch emits ordinary bottom pieces (index, offset) in the MRCA, the
        // others emit positional pieces (forward start in the parent + its top-parse hint).
        const size_t nUp = P.up.size() - 1; // number of hops
        if (nUp >= 2 && nUp <= (size_t)MAX_CHAIN && !P.levelSyncUp) {
            // the whole up phase depth first in one launch (no intermediate frontiers)
            UpTables<C> tabs;
            for (int k = 0; k < MAX_CHAIN; ++k) {
                const size_t kk = (size_t)k < nUp ? (size_t)k : 0;
                const DeviceGenome &UG = D.genomes[(size_t)P.up[kk]];
                tabs.t[k] = (const ChainRec<C> *)(kk + 1 == nUp ? UG.chainLast : UG.chainMid);
            }
            tabs.n = (int)nUp;
            P.timer.begin("k_up_chain", s, launch);
            hipLaunchKernelGGL((k_up_chain<C>), dim3(residentGrid(k_up_chain<C>, upChainLdsBytes((int)nUp))), dim3(256), upChainLdsBytes((int)nUp), s, tabs, P.frontier(cur), inCnt(), cap, P.frontier(cur ^ 1),
                               outCnt(), minLen, cnt, kstat());
            P.timer.end(s);
            ++launch;
            cur ^= 1;
            level = outLevel++;
        } else {
            {
                const bool last = nUp == 1;
                P.timer.begin("k_up_first", s, launch);
                hipLaunchKernelGGL((k_up_first<C>), dim3(GRID), dim3(256), 0, s, (const UpRec<C> *)D.genomes[(size_t)P.src].up,
                                   P.frontier(cur), inCnt(), cap, P.frontier(cur ^ 1), outCnt(), minLen, (int)last, cnt, kstat());
                P.timer.end(s);
                ++launch;
                cur ^= 1;
                level = outLevel++;
            }
            for (size_t k = 1; k < nUp; ++k) {
                const bool last = k + 1 == nUp;
                P.timer.begin("k_up_walk", s, launch);
                hipLaunchKernelGGL((k_up_walk<C>), dim3(GRID), dim3(256), 0, s, (const UpRec<C> *)D.genomes[(size_t)P.up[k]].up,
                                   P.frontier(cur), inCnt(), cap, P.frontier(cur ^ 1), outCnt(), minLen, (int)last, cnt, kstat());
                P.timer.end(s);
                ++launch;
                cur ^= 1;
                level = outLevel++;
            }
        }
        curGenome = P.mrca;
        curTop = false;
    }
    // table builder: the run stops at the captured frontier, which stays in the plan's buffers
    auto endRun = [&]() {
        HIP_OK(hipEventRecord(P.evEnd, s));
        const int words = STAT_LAUNCH0 + 2 * std::min(launch + 1, (int)MAX_LAUNCHES);
        hipLaunchKernelGGL(k_fold_stats, dim3(1), dim3(STAT_PITCH), 0, s, cnt, words);
        // one readback: the scalar slots, the per-launch statistics and the record total; the 64 KB of frontier counters
        // only when somebody needs them (the table builder, or below after an overflow)
        if (P.captureUp || P.captureFinal) {
            HIP_OK(hipMemcpyAsync(P.pinned, cnt, 8 * CNT_SLOTS, hipMemcpyDeviceToHost, s));
        } else {
            HIP_OK(hipMemcpyAsync(P.pinned, cnt, 8 * CNT_FRONT0, hipMemcpyDeviceToHost, s));
            HIP_OK(hipMemcpyAsync(P.pinned + CNT_KSTAT0, cnt + CNT_KSTAT0, 8 * (size_t)(words - STAT_LAUNCH0), hipMemcpyDeviceToHost, s));
        }
        HIP_OK(hipMemcpyAsync(P.pinned + CNT_SLOTS, P.total.p, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (P.pinned[CNT_OVERFLOW] && !(P.captureUp || P.captureFinal)) { // the retry sizes the buffers from the frontier counters
            HIP_OK(hipMemcpyAsync(P.pinned, cnt, 8 * CNT_SLOTS, hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
        }
        memcpy(hostCounters, P.pinned, 8 * CNT_SLOTS);
        const size_t usedStats = (size_t)(words - STAT_LAUNCH0); // launches this run did not make count nothing
        memset(hostCounters + CNT_KSTAT0 + usedStats, 0, 8 * (2 * (size_t)MAX_LAUNCHES - usedStats));
    };
    auto capture = [&]() {
        P.capturedBuf = cur;
        P.capturedLevel = level;
        HIP_OK(hipEventRecord(P.evWalk, s));
        endRun();
    };
    if (P.captureUp) {
        capture();
        return;
    }
    if (!P.climb.empty() && !through) {
        // mapRecursiveParalogies (halSegmentMapper.cpp:525-576), coalescenceLimit above the MRCA: at every genome c_i from
        // the MRCA up to the child of the limit, the pieces (walked through c_i's top tiling) are expanded to their paralogy
        // rings (mapSelf) and the ring members are mapped back DOWN to the MRCA without dupes; the pieces themselves (not
        // their paralogs) go up one more level (mapUp, always with dupes).  The union R of what came back down (top pieces
        // in the MRCA) replaces the frontier.  Exact duplicates the reference removes with sort+unique at every level
        // are removed here by the set semantics of the finishing step.
        // Buffers: F = pieces in c_i, T = F on the top tiling, N = pieces in c_{i+1}, A/B = the way down, R = result.
        const int rBuf = cur ^ 1;
        int rot[3] = {cur, 2, 3}; // F, T, N
        const int aBuf = 4, bBuf = 5;
        int nextLevel = outLevel;
        auto cntOf = [&](int lv) { return cnt + CNT_FRONT0 + (size_t)lv; };
        const int rLevel = nextLevel++;
        int fLevel = level;
        bool fTop = curTop;
        for (size_t i = 0; i + 1 < P.climb.size(); ++i) {
            const int g = P.climb[i];
            const DeviceGenome &G = D.genomes[(size_t)g];
            const int fBuf = rot[0], tBuf = rot[1], nBuf = rot[2];
            int tLevel = fLevel, tb = fBuf;
            if (!fTop) {
                tLevel = nextLevel++;
                tb = tBuf;
                P.timer.begin("k_parse_up", s, launch);
                hipLaunchKernelGGL((k_parse_up<C>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)G.bot, (const TopRec<C> *)G.top,
                                   P.frontier(fBuf), cntOf(fLevel), cap, P.frontier(tb), cntOf(tLevel), cnt, kstat());
                P.timer.end(s);
                ++launch;
            }
            // ring members of c_i; for i == 0 they already live in the MRCA
            const int pLevel = i == 0 ? rLevel : nextLevel++;
            const int pb = i == 0 ? rBuf : aBuf;
            P.timer.begin("k_ring", s, launch);
            hipLaunchKernelGGL((k_ring<C>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)G.top, P.frontier(tb), cntOf(tLevel), cap,
                               P.frontier(pb), cntOf(pLevel), minLen, cnt, kstat());
            P.timer.end(s);
            ++launch;
            // the pieces themselves one level up (before the buffers of the way down are reused)
            int nLevel = -1;
            if (i + 2 < P.climb.size()) {
                nLevel = nextLevel++;
                P.timer.begin("k_up_first", s, launch);
                hipLaunchKernelGGL((k_up_first<C>), dim3(GRID), dim3(256), 0, s, (const UpRec<C> *)G.up, P.frontier(tb), cntOf(tLevel), cap,
                                   P.frontier(nBuf), cntOf(nLevel), minLen, 1, cnt, kstat());
                P.timer.end(s);
                ++launch;
            }
            // ring members back down to the MRCA: c_i -> c_{i-1} -> ... -> c_0, no dupes (mapRecursiveDown(..., false, ...))
            int dBuf = pb, dLevel = pLevel;
            for (size_t j = i; j > 0; --j) {
                const int pg = P.climb[j], cg = P.climb[j - 1];
                const DeviceGenome &PG = D.genomes[(size_t)pg];
                const DeviceGenome &CG = D.genomes[(size_t)cg];
                const int slot = P.h->img.genomes[(size_t)pg].childSlotOf(cg);
                const int xBuf = dBuf == aBuf ? bBuf : aBuf;
                const int xLevel = nextLevel++;
                P.timer.begin("k_parse_down", s, launch);
                hipLaunchKernelGGL((k_parse_down<C>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)PG.top, (const BotRec<C> *)PG.bot,
                                   P.frontier(dBuf), cntOf(dLevel), cap, P.frontier(xBuf), cntOf(xLevel), cnt, kstat());
                P.timer.end(s);
                ++launch;
                const bool intoR = j == 1;
                const int yBuf = intoR ? rBuf : (xBuf == aBuf ? bBuf : aBuf);
                const int yLevel = intoR ? rLevel : nextLevel++;
                P.timer.begin("k_down_ring", s, launch);
                hipLaunchKernelGGL((k_down_ring<C, false>), dim3(GRID), dim3(256), 0, s, (const DownRec<C> *)PG.downRec[(size_t)slot],
                                   (const TopRec<C> *)CG.top, P.frontier(xBuf), cntOf(xLevel), cap, P.frontier(yBuf), cntOf(yLevel), minLen,
                                   0, cnt, kstat(), (uint32_t *)nullptr);
                P.timer.end(s);
                ++launch;
                dBuf = yBuf;
                dLevel = yLevel;
            }
            if (nLevel < 0)
                break;
            // next level: N becomes F; the old F and T buffers are free again
            rot[0] = nBuf;
            rot[1] = fBuf;
            rot[2] = tBuf;
            fLevel = nLevel;
            fTop = false;
        }
        cur = rBuf;
        level = rLevel;
        outLevel = nextLevel;
        curTop = true;
        curGenome = P.mrca;
    }
    bool finalized = through; // the last walk kernel already produced final pieces and the per-interval counts
    if (P.tgt != P.mrca && !through) {
        if (curTop) { // source is the MRCA itself and is walked through its top tiling
            const DeviceGenome &G = D.genomes[(size_t)curGenome];
            P.timer.begin("k_parse_down", s, launch);
            hipLaunchKernelGGL((k_parse_down<C>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)G.top, (const BotRec<C> *)G.bot,
                               P.frontier(cur), inCnt(), cap, P.frontier(cur ^ 1), outCnt(), cnt, kstat());
            P.timer.end(s);
            ++launch;
            cur ^= 1;
            level = outLevel++;
            curTop = false;
        }
        for (size_t k = 0; k < P.down.size(); ++k) {
            const int parent = P.down[k].first, slot = P.down[k].second;
            const int child = P.h->img.genomes[(size_t)parent].children[(size_t)slot];
            const DeviceGenome &PG = D.genomes[(size_t)parent];
            const DeviceGenome &CG = D.genomes[(size_t)child];
            P.timer.begin("k_down_ring", s, launch);
            if (child == P.tgt) {
                hipLaunchKernelGGL((k_down_ring<C, true>), dim3(GRID), dim3(256), 0, s, (const DownRec<C> *)PG.downRec[(size_t)slot],
                                   (const TopRec<C> *)CG.top, P.frontier(cur), inCnt(), cap, P.frontier(cur ^ 1), outCnt(), minLen,
                                   (int)(P.opts.traverse_dupes != 0), cnt, kstat(), (uint32_t *)P.perQuery.p);
                finalized = true;
            } else {
                hipLaunchKernelGGL((k_down_ring<C, false>), dim3(GRID), dim3(256), 0, s, (const DownRec<C> *)PG.downRec[(size_t)slot],
                                   (const TopRec<C> *)CG.top, P.frontier(cur), inCnt(), cap, P.frontier(cur ^ 1), outCnt(), minLen,
                                   (int)(P.opts.traverse_dupes != 0), cnt, kstat(), (uint32_t *)nullptr);
            }
            P.timer.end(s);
            ++launch;
            cur ^= 1;
            level = outLevel++;
            curTop = true;
            curGenome = child;
            if (child != P.tgt) {
                P.timer.begin("k_parse_down", s, launch);
                hipLaunchKernelGGL((k_parse_down<C>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)CG.top,
                                   (const BotRec<C> *)CG.bot, P.frontier(cur), inCnt(), cap, P.frontier(cur ^ 1), outCnt(), cnt,
                                   kstat());
                P.timer.end(s);
                ++launch;
                cur ^= 1;
                level = outLevel++;
                curTop = false;
            }
        }
    }
    if (P.captureFinal) {
        if (!finalized)
            throw std::runtime_error("internal: captureFinal on a path without a final down hop");
        capture();
        return;
    }
    // final pieces live in the target genome
    const DeviceGenome &TG = D.genomes[(size_t)P.tgt];
    if (!finalized) {
        P.timer.begin("k_finalize", s, launch);
        if (curTop)
            hipLaunchKernelGGL((k_finalize<TopRec<C>>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)TG.top, P.frontier(cur), inCnt(),
                               cap, P.mapped(0), (uint32_t *)P.perQuery.p, cnt, kstat(), 1);
        else
            hipLaunchKernelGGL((k_finalize<BotRec<C>>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)TG.bot, P.frontier(cur), inCnt(),
                               cap, P.mapped(0), (uint32_t *)P.perQuery.p, cnt, kstat(), 0);
        P.timer.end(s);
    }
    HIP_OK(hipEventRecord(P.evWalk, s));

    if (!through) { // (the whole-path table kernel wrote its pieces grouped, with offset[] and perQuery[])
        exclusiveScan(P, (const uint32_t *)P.perQuery.p, nq, (uint32_t *)P.offset.p, (uint32_t *)P.total.p, s);
        P.timer.begin("k_scatter", s);
        if (finalized)
            hipLaunchKernelGGL(k_scatter_front, dim3(GRID), dim3(256), 0, s, P.frontier(cur), inCnt(), cap, (const uint32_t *)P.offset.p,
                               (uint32_t *)P.cursor.p, P.mapped(1), cnt);
        else
            hipLaunchKernelGGL(k_scatter, dim3(GRID), dim3(256), 0, s, P.mapped(0), cnt + CNT_MAPPED, cap, (const uint32_t *)P.offset.p,
                               (uint32_t *)P.cursor.p, P.mapped(1));
        P.timer.end(s);
    }
    // finishing: register-resident fast path per size class (each kernel picks the intervals of its class),
    // general LDS path for the rest
    // classLists: [general | 9-16 | 17-32 | 33-64 pieces], nq entries each; classCounts: [general, 3 classes]
    // (splitting every list into sub-lists with their counters on separate cache lines was measured: no gain — the blocks of
    // k_finish_fast<C, 8> reach their bulk appends at different times, unlike the statistics all wavefronts add at their end)
    uint32_t *generalList = (uint32_t *)P.classLists.p;
    unsigned long long *generalCount = (unsigned long long *)P.classCounts.p;
    uint32_t *classLists = generalList + nq;
    unsigned long long *classCounts = generalCount + 1;
    HIP_OK(hipMemsetAsync(generalCount, 0, 32, s));
    const int blocks = P.opts.emit_blocks;
    if (blocks) {
        hipLaunchKernelGGL(k_all_general, dim3(GRID), dim3(256), 0, s, (const uint32_t *)P.perQuery.p, nq, (uint32_t *)P.nOut.p, generalList,
                           generalCount);
    } else {
        P.timer.begin("k_finish_fast", s);
        const int gridG8 = (int)std::max<uint32_t>((uint32_t)GRID, (nq + FAST_LIST_CAP - 1) / FAST_LIST_CAP); // a block's share fits its LDS lists
#define HGX_FAST(G)                                                                                                    \
        hipLaunchKernelGGL((k_finish_fast<C, G>), dim3(G == 8 ? gridG8 : GRID), dim3(256), 0, s, P.mapped(1), (const uint32_t *)P.offset.p,  \
                           (const uint32_t *)P.perQuery.p, nq, (const int64_t *)TG.seqStart, (int)TG.numSeq,               \
                           (hgx_record *)P.grouped.p, (uint32_t *)P.nOut.p, generalList, generalCount, classLists, classCounts)
        HGX_FAST(8);
        HGX_FAST(16);
        HGX_FAST(32);
        HGX_FAST(64);
#undef HGX_FAST
        P.timer.end(s);
    }
    P.timer.begin("k_finish_lds", s);
    hipLaunchKernelGGL((k_finish_lds<C, 256>), dim3(std::min<uint32_t>(std::max<uint32_t>(nq, 1), 1u << 14)), dim3(64), 0, s, P.mapped(1),
                       (const uint32_t *)P.offset.p, (const uint32_t *)P.perQuery.p, (const uint32_t *)generalList,
                       (const unsigned long long *)generalCount, (const int64_t *)TG.seqStart, (int)TG.numSeq,
                       (hgx_record *)P.grouped.p, (uint32_t *)P.nOut.p, (uint32_t *)P.deferredList.p, (uint32_t *)P.needCap.p, cnt, blocks);
    P.timer.end(s);
    // The common case has no deferred interval: number the output and compact it right away, and read the counters and
    // the record total back with ONE synchronisation.  runPlan redoes the tail when an interval was deferred.
    exclusiveScan(P, (const uint32_t *)P.nOut.p, nq, (uint32_t *)P.outOffset.p, (uint32_t *)P.total.p, s);
    P.timer.begin("k_compact_records", s);
    hipLaunchKernelGGL(k_compact_records, dim3(GRID), dim3(256), 0, s, (const hgx_record *)P.grouped.p, (const uint32_t *)P.offset.p,
                       (const hgx_record *)nullptr, (const int32_t *)nullptr, 0, (const uint32_t *)P.nOut.p, (const uint32_t *)P.outOffset.p, nq,
                       (hgx_record *)P.outRecords.p);
    P.timer.end(s);
    endRun();
}

__global__ void k_fill_big_slot(const uint32_t *deferredList, uint32_t nd, int32_t *bigSlot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nd)
        bigSlot[deferredList[i]] = (int32_t)i;
}

template <typename C>
static void runPlan(hgx_liftover_plan &P, size_t n, const int64_t *dS, const int64_t *dE, const uint8_t *dStrand, hipStream_t s,
                    const hgx_record **dOut, size_t *nOut) {
    if (n > P.maxQueries)
        throw std::runtime_error("batch larger than the plan's max_queries");
    if (n >= ((size_t)1 << 31))
        throw std::runtime_error("batch of 2^31 or more intervals; split it");
    std::vector<unsigned long long> hcv(CNT_SLOTS);
    unsigned long long *hc = hcv.data();
    const DeviceImage &D = *P.h->dev;
    HIP_OK(hipSetDevice(D.device));
    P.stats = hgx_liftover_stats{};
    if (n == 0) {
        *dOut = (const hgx_record *)P.outRecords.p;
        *nOut = 0;
        return;
    }
    bool builtNow = false;
    if (!P.composed && P.composedAfter != ~0ull) {
        // the batch that takes the plan past its threshold is already served from the table
        if (P.walked + n >= P.composedAfter) {
            g_phases.start();
            P.composed = ensureComposed(P.h, P.src, P.composedThrough ? P.tgt : P.mrca, P.composedThrough, P.opts, !P.opts.emit_blocks);
            g_phases.lap("merged: workspaces released");
            builtNow = true;
        } else {
            P.walked += n;
        }
    }
    P.timer.beginRun();
    P.mergedOffThisRun = false;
    for (;;) {
        runOnce<C>(P, n, dS, dE, dStrand, s, hc);
        if (!hc[CNT_OVERFLOW]) {
            const bool merged = P.composed && P.composed->mRecs && !P.opts.emit_blocks && !P.mergedOffThisRun;
            if (merged && hc[CNT_DEFERRED] && (hc[CNT_BIGFAIL] || hc[CNT_DEFERRED] > P.liftBigSlots)) {
                // intervals outgrew the LDS finishing kernel and the scratch area of k_finish_big was too small (or not there
                // yet): size it from what this run needed and repeat the batch
                const unsigned long long slots = std::max<unsigned long long>(64, 2 * hc[CNT_DEFERRED]);
                const unsigned long long pieces = std::max<unsigned long long>(512, 2 * hc[CNT_MAXNEED]);
                if (pieces > (1ull << 26) || slots * pieces >= (1ull << 31))
                    P.mergedOffThisRun = true; // (the multi-kernel path sizes its scratch per run)
                else {
                    P.liftBigSlots = (uint32_t)std::max<unsigned long long>(P.liftBigSlots, slots);
                    P.liftBigCap = (int)std::max<unsigned long long>((unsigned long long)P.liftBigCap, pieces);
                    P.allocate(P.cap);
                }
                P.timer.dropRun();
                continue;
            }
            if (merged && P.liftRestCount && P.liftRestSkipped) { // intervals were passed on to launches this run did not make
                P.liftRestSeen = true;
                P.timer.dropRun();
                continue;
            }
            if (merged && P.liftRestCount)
                P.liftRestSeen = true;
            break;
        }
        // a frontier outgrew the workspace: size it from the largest count seen and run again
        unsigned long long need = 0;
        for (int lv = 0; lv < MAX_LEVELS; ++lv) {
            unsigned long long tot = 0, mx = 0;
            for (int sgm = 0; sgm < NSEG; ++sgm) {
                tot += hc[CNT_FRONT0 + sgm * SEG_PITCH + lv];
                mx = std::max(mx, hc[CNT_FRONT0 + sgm * SEG_PITCH + lv]);
            }
            need = std::max(need, std::max(tot, mx * NSEG)); // the fullest segment sets the capacity
        }
        need = std::max(need, hc[CNT_LIFT_TOTAL]); // (the single-pass kernel's output)
        need = std::max<unsigned long long>(need + need / 4, 2ull * P.cap);
        if (need >= (1ull << 32))
            throw std::runtime_error("liftover batch expands to more than 2^32 pieces; submit smaller batches");
        P.timer.dropRun();
        P.allocate((uint32_t)need);
    }
    const uint32_t nq = (uint32_t)n;
    const DeviceGenome &TG = D.genomes[(size_t)P.tgt];
    unsigned long long *cnt = (unsigned long long *)P.counters.p;
    const bool mergedRun = P.composed && P.composed->mRecs && !P.opts.emit_blocks && !P.mergedOffThisRun;
    // (a single-pass run has finished its deferred intervals itself and placed their records)
    const uint32_t nDeferredSeen = (uint32_t)hc[CNT_DEFERRED];
    const uint32_t nDef = mergedRun ? 0u : nDeferredSeen;
    int bigCap = 0;
    if (nDef > 0) {
        bigCap = (int)std::max<unsigned long long>(512, 2 * hc[CNT_MAXNEED]);
        HIP_OK(hipMemsetAsync(P.bigSlot.p, 0xFF, 4 * (n + 1), s)); // -1: not deferred
        hipLaunchKernelGGL(k_fill_big_slot, dim3((nDef + 255) / 256), dim3(256), 0, s, (const uint32_t *)P.deferredList.p, nDef,
                           (int32_t *)P.bigSlot.p);
        for (;;) {
            const size_t slice = finishSliceBytes<C>(bigCap);
            P.scratch.ensure(slice * nDef);
            P.bigRecords.ensure(sizeof(hgx_record) * (size_t)bigCap * nDef);
            unsigned long long zero[2] = {0, 0};
            HIP_OK(hipMemcpyAsync(cnt + CNT_MAXNEED, zero, 16, hipMemcpyHostToDevice, s)); // MAXNEED, BIGFAIL
            P.timer.begin("k_finish_big", s);
            const int bigInLds = bigLdsOk<C>(slice) ? 1 : 0;
            hipLaunchKernelGGL((k_finish_big<C>), dim3(nDef), dim3(64), bigInLds ? slice : 0, s, P.mapped(1), (const uint32_t *)P.offset.p,
                               (const uint32_t *)P.perQuery.p, (const uint32_t *)P.deferredList.p, nDef, bigCap,
                               (unsigned char *)P.scratch.p, slice, (const int64_t *)TG.seqStart, (int)TG.numSeq,
                               (hgx_record *)P.bigRecords.p, (uint32_t *)P.nOut.p, cnt, P.opts.emit_blocks, 0, (uint32_t *)nullptr, 0u,
                               (uint32_t *)nullptr, (unsigned long long *)nullptr, bigInLds);
            P.timer.end(s);
            unsigned long long r[2];
            HIP_OK(hipMemcpyAsync(r, cnt + CNT_MAXNEED, 16, hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            if (!r[1])
                break;
            bigCap = (int)std::max<unsigned long long>(2ull * bigCap, 2 * r[0]);
            if (bigCap > (1 << 26))
                throw std::runtime_error("an interval maps to more than 2^26 pieces; not supported");
        }
    }
    uint32_t totalRecords = (uint32_t)(P.pinned[CNT_SLOTS] & 0xFFFFFFFFull);
    if (nDef > 0) {
        // the deferred intervals' records live in bigRecords: number and compact again
        exclusiveScan(P, (const uint32_t *)P.nOut.p, nq, (uint32_t *)P.outOffset.p, (uint32_t *)P.total.p, s);
        HIP_OK(hipMemcpyAsync(&totalRecords, P.total.p, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        P.outRecords.ensure(sizeof(hgx_record) * std::max<size_t>(totalRecords, 1));
        P.timer.begin("k_compact_records", s);
        hipLaunchKernelGGL(k_compact_records, dim3(GRID), dim3(256), 0, s, (const hgx_record *)P.grouped.p, (const uint32_t *)P.offset.p,
                           (const hgx_record *)P.bigRecords.p, (const int32_t *)P.bigSlot.p, bigCap, (const uint32_t *)P.nOut.p,
                           (const uint32_t *)P.outOffset.p, nq, (hgx_record *)P.outRecords.p);
        P.timer.end(s);
        HIP_OK(hipEventRecord(P.evEnd, s));
        HIP_OK(hipStreamSynchronize(s));
    }
    P.timer.endRun(hc);
    unsigned long long topAll = 0, botAll = 0;
    for (int k = 0; k < MAX_LAUNCHES; ++k) {
        topAll += hc[CNT_KSTAT0 + 2 * k];
        botAll += hc[CNT_KSTAT0 + 2 * k + 1];
    }
    float walk = 0, tot = 0;
    if (!(mergedRun && P.timer.mode == 0)) { // (a single-pass run without kernel events records none of its own either)
        HIP_OK(hipEventElapsedTime(&walk, P.evStart, P.evWalk));
        HIP_OK(hipEventElapsedTime(&tot, P.evStart, P.evEnd));
    }
    if (builtNow)
        g_phases.lap("first batch from the table");
    P.stats.queries = n;
    P.stats.source_pieces = hc[CNT_SRC_PIECES];
    P.stats.top_derefs = topAll;
    P.stats.bottom_derefs = botAll;
    P.stats.mapped_pieces = hc[CNT_MAPPED];
    P.stats.records = totalRecords;
    P.stats.deferred_queries = nDeferredSeen;
    P.stats.walk_ms = walk;
    P.stats.total_ms = tot;
    P.stats.composed_records = P.composed ? P.composed->numRecs : 0;
    P.stats.composed_build_ms = P.composed ? P.composed->buildMs : 0;
    P.stats.composed_kind = P.composed ? (P.composed->through ? 2 : 1) : 0;
    if (P.composed && P.composed->mRecs && !P.opts.emit_blocks && !P.mergedOffThisRun) {
        P.stats.composed_kind = 3;
        P.stats.composed_records = P.composed->mNum;
        P.stats.composed_build_ms = P.composed->buildMs + P.composed->mBuildMs;
        P.stats.general_queries = P.generalQueries;
        P.stats.composed_flagged = P.composed->mFlagged;
    }
    *dOut = (const hgx_record *)P.outRecords.p;
    *nOut = totalRecords;
}

static void runHostArrays(hgx_liftover_plan *P, const std::vector<int64_t> &gs, const std::vector<int64_t> &ge, const std::vector<uint8_t> &st,
                          std::vector<hgx_record> &out);

hgx_liftover_plan *createLiftoverPlan(hgx_alignment *h, int src, int tgt, const hgx_liftover_opts &opts, size_t maxQueries,
                                      bool allowComposed, bool tableBuilder) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); liftover needs the HIP path");
    const Image &img = h->img;
    if (src < 0 || tgt < 0 || src >= (int)img.genomes.size() || tgt >= (int)img.genomes.size())
        throw std::runtime_error("genome id out of range");
    std::unique_ptr<hgx_liftover_plan> P(new hgx_liftover_plan);
    P->h = h;
    P->tableBuilder = tableBuilder;
    P->device = h->dev->device;
    P->src = src;
    P->tgt = tgt;
    P->opts = opts;
    P->mrca = img.lca(src, tgt);
    if (opts.coalescence_limit >= (int)img.genomes.size())
        throw std::runtime_error("coalescenceLimit: no such genome");
    if (opts.coalescence_limit >= 0 && opts.coalescence_limit != P->mrca && opts.traverse_dupes) {
        // mapSource (halSegmentMapper.cpp:616-621): the paralogy phase runs only with dupes on.  The limit has to be an
        // ancestor of the MRCA; the reference finds out while climbing (":541 Hit root genome ..."), here it is checked up front.
        for (int g = P->mrca;; g = img.genomes[(size_t)g].parent) {
            if (g < 0)
                throw std::runtime_error("Hit root genome when attempting to map paralogies");
            P->climb.push_back(g);
            if (g == opts.coalescence_limit)
                break;
        }
        P->numFrontiers = 6;
    }
    for (int g = src;; g = img.genomes[(size_t)g].parent) {
        P->up.push_back(g);
        if (g == P->mrca)
            break;
    }
    {
        const size_t nUp = P->up.size() - 1;
        if (nUp >= 2 && nUp <= (size_t)MAX_CHAIN && !P->levelSyncUp)
            for (size_t k = 0; k < nUp; ++k)
                ensureChainTables(img, *h->dev, P->up[k], k + 1 < nUp, k + 1 == nUp);
    }
    for (size_t j = 1; j < P->climb.size(); ++j) // the paralogy phase maps ring members back down the climb path
        ensureDownTable(img, *h->dev, P->climb[j], img.genomes[(size_t)P->climb[j]].childSlotOf(P->climb[j - 1]));
    std::vector<int> chain; // tgt ... mrca
    for (int g = tgt; g != P->mrca; g = img.genomes[(size_t)g].parent)
        chain.push_back(g);
    int parent = P->mrca;
    for (size_t k = chain.size(); k-- > 0;) {
        // mapRecursiveDown picks the first child that is the target or lies on the path to it
        // (halSegmentMapper.cpp:208-219); on a tree that is the unique child towards the target
        const int slot = img.genomes[(size_t)parent].childSlotOf(chain[k]);
        P->down.emplace_back(parent, slot);
        ensureDownTable(img, *h->dev, parent, slot);
        parent = chain[k];
    }
    const size_t climbHops = P->climb.empty() ? 0 : P->climb.size() - 1;
    const size_t climbLaunches = 3 * climbHops + climbHops * climbHops; // parse-up, ring, up + two per level of the way back down
    if ((int)(P->up.size() + 2 * P->down.size() + climbLaunches) + 4 >= MAX_LEVELS - 1 ||
        (int)(P->up.size() + 2 * P->down.size() + climbLaunches) + 4 >= MAX_LAUNCHES)
        throw std::runtime_error("tree path between the genomes is too long for the counter block");
    // BlockLiftover::visitBegin (halBlockLiftover.cpp:24-30): walk the source through its top tiling when it has one
    P->srcTop = img.genomes[(size_t)src].numTop > 0;
    // BlockMapper::map (halBlockMapper.cpp:79-86) chooses differently: bottom segments iff the reference genome is the MRCA
    // and not the query genome itself, top segments otherwise
    if (opts.block_mapper_source == 1)
        P->srcTop = !(P->mrca == src && src != tgt);
    else if (opts.block_mapper_source == 2 || opts.block_mapper_source == 3) // the caller names the tiling (hgx_blockviz.cpp)
        P->srcTop = opts.block_mapper_source == 2;
    // (a genome without the tiling its rule names — the root walked through its top segments by BlockMapper's rule for a
    // self-alignment — maps nothing; a caller that names a tiling the genome does not have is mistaken)
    if (opts.block_mapper_source >= 2 && (P->srcTop ? img.genomes[(size_t)src].numTop <= 0 : img.genomes[(size_t)src].numBot <= 0))
        throw std::runtime_error("the source genome has no segments of the tiling the walk was asked to start from");
    ensureLocateTable(img, *h->dev, src, P->srcTop ? 0 : 1);
    P->maxQueries = std::max<size_t>(maxQueries, 1);
    HIP_OK(hipSetDevice(h->dev->device));
    HIP_OK(hipEventCreate(&P->evStart));
    HIP_OK(hipEventCreate(&P->evWalk));
    HIP_OK(hipEventCreate(&P->evEnd));
    // pieces per interval the workspace starts with (grown on demand); the table builder's intervals are single source
    // segments, which yield a few pieces each, and its batch is the whole genome
    unsigned long long perQuery = allowComposed ? 16ull : 4ull;
    // A plan that has walked a few times as many intervals as the source genome has segments switches to a composed table
    // (runPlan): building one costs about as much as walking one interval per source segment, a table lookup a third of a
    // walk.  The table of the whole path when the target lies below the MRCA, the up table otherwise.
    // HGX_COMPOSED_UP=1 builds it right away, =0 forbids it; HGX_COMPOSED_THROUGH=0 keeps to the up table.
    if (allowComposed && src != P->mrca && P->srcTop && opts.min_length == 0) {
        const char *e = getenv("HGX_COMPOSED_UP");
        const bool force = e && e[0] == '1', forbid = e && e[0] == '0';
        const char *t = getenv("HGX_COMPOSED_THROUGH");
        P->composedThrough = tgt != P->mrca && !(t && t[0] == '0');
        // A table costs about as much as walking one interval per source segment, and an interval covers a few segments:
        // the plan changes over with the batch that brings its intervals to a quarter of the source's segments
        // (HGX_COMPOSED_AFTER=<x>: to x times the source's segments).
        double after = 0.25;
        if (const char *a = getenv("HGX_COMPOSED_AFTER"))
            after = atof(a);
        P->composedAfter = forbid ? ~0ull : (force ? 0ull : (unsigned long long)std::max(1.0, after * (double)img.genomes[(size_t)src].numTop));
        // a plan whose first full batch already changes it over to a whole-path table holds output lines, not walked pieces
        if (P->composedThrough && P->composedAfter <= P->maxQueries)
            perQuery = 4ull;
        if (force)
            P->composed = ensureComposed(h, src, P->composedThrough ? tgt : P->mrca, P->composedThrough, opts, !opts.emit_blocks);
    }
    const unsigned long long want = std::max<unsigned long long>(1ull << 16, perQuery * P->maxQueries);
    P->allocate((uint32_t)std::min<unsigned long long>(want, (1ull << 32) - 2));
    return P.release();
}

// Builds (once per alignment and pair) a composed table.  Up table (src -> mrca): every source top segment is lifted to the
// MRCA as one interval by the ordinary walk (a plan with captureUp).  Table of the whole path (src -> tgt, through): the
// same batch runs the complete walk with the plan's own options (dupes, coalescenceLimit) and the FINAL pieces are kept.
// The pieces are sorted by source position.  Serialised; the table lives as long as the device image.
template <typename C>
static void buildComposed(hgx_alignment *h, int src, int dst, bool through, const hgx_liftover_opts &opts, ComposedUp &out) {
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { g_phases.lap((std::string("pieces: ") + what).c_str()); };
    const GenomeTables &S = h->img.genomes[(size_t)src];
    const DeviceImage &D = *h->dev;
    const size_t nt = (size_t)S.numTop;
    hgx_liftover_opts o{};
    o.traverse_dupes = 1;
    o.coalescence_limit = -1;
    if (through) {
        o.traverse_dupes = opts.traverse_dupes;
        o.coalescence_limit = opts.coalescence_limit;
        o.block_mapper_source = opts.block_mapper_source;
    }
    std::unique_ptr<hgx_liftover_plan, void (*)(hgx_liftover_plan *)> P(createLiftoverPlan(h, src, dst, o, nt, /*allowComposed=*/false, /*tableBuilder=*/true),
                                                                       destroyLiftoverPlan);
    (through ? P->captureFinal : P->captureUp) = true;
    P->timer.mode = 0;
    lap("builder plan");
    HIP_OK(hipSetDevice(D.device));
    hipStream_t s = nullptr;
    // every source top segment as one forward interval, walked to the capture point
    DevBuf dS, dE, dT;
    dS.ensure(8 * std::max<size_t>(nt, 1));
    dE.ensure(8 * std::max<size_t>(nt, 1));
    dT.ensure(std::max<size_t>(nt, 1));
    if (nt)
        hipLaunchKernelGGL((k_table_queries<C>), dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, (const TopRec<C> *)D.genomes[(size_t)src].top,
                           (uint32_t)nt, (int64_t *)dS.p, (int64_t *)dE.p, (uint8_t *)dT.p);
    const hgx_record *ignoredRecords = nullptr;
    size_t ignoredCount = 0;
    runLiftoverPlan(P.get(), nt, (const int64_t *)dS.p, (const int64_t *)dE.p, (const uint8_t *)dT.p, s, &ignoredRecords, &ignoredCount);
    lap("walk of every source segment");
    size_t n = 0;
    if (nt) {
        if (P->capturedBuf < 0)
            throw std::runtime_error("internal: the table builder's run did not reach its capture point");
        const uint32_t segCap = P->cap / NSEG;
        for (int sg = 0; sg < NSEG; ++sg)
            n += (size_t)std::min<unsigned long long>(P->pinned[CNT_FRONT0 + (size_t)sg * SEG_PITCH + (size_t)P->capturedLevel], segCap);
    }
    if (n >= ((size_t)1 << 32) - 1)
        throw std::runtime_error("composed table too large");
    // ~4 records per bucket
    int64_t buckets = 1;
    while (buckets < (int64_t)n / 4 && buckets < ((int64_t)1 << 22))
        buckets <<= 1;
    int shift = 0;
    while (((S.totalLength - 1) >> shift) >= buckets)
        ++shift;
    const uint32_t nb = (uint32_t)(((S.totalLength - 1) >> shift) + 1);
    const size_t nAlloc = std::max<size_t>(n, 1);
    HIP_OK(hipMalloc(&out.recs, nAlloc * sizeof(ComposedRec<C>)));
    HIP_OK(hipMemsetAsync(out.recs, 0, nAlloc * sizeof(ComposedRec<C>), s));
    HIP_OK(hipMalloc((void **)&out.coarse, ((size_t)nb + 1) * 4));
    HIP_OK(hipMemsetAsync(out.coarse, 0xFF, ((size_t)nb + 1) * 4, s));
    HIP_OK(hipMalloc((void **)&out.starts, ((size_t)nb + 1) * 4));
    size_t bytes = nAlloc * sizeof(ComposedRec<C>) + ((size_t)nb + 1) * 8;
    if (!through) {
        HIP_OK(hipMalloc(&out.eo, nAlloc * sizeof(C)));
        bytes += nAlloc * sizeof(C);
    }
    if (n) {
        const Frontier F = P->frontier(P->capturedBuf);
        const unsigned long long *levelCount = (const unsigned long long *)P->counters.p + CNT_FRONT0 + (size_t)P->capturedLevel;
        DevBuf keys[2], slots[2], tmp, bad;
        for (int k = 0; k < 2; ++k) {
            keys[k].ensure(8 * n);
            slots[k].ensure(4 * n);
        }
        bad.ensure(8);
        HIP_OK(hipMemsetAsync(bad.p, 0, 8, s));
        hipLaunchKernelGGL(k_table_keys, dim3(GRID), dim3(256), 0, s, F, levelCount, P->cap, (uint64_t *)keys[0].p, (uint32_t *)slots[0].p);
        int endBit = 1;
        while (endBit < 64 && (S.totalLength >> endBit) != 0)
            ++endBit;
        size_t tmpBytes = 0;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, (const uint64_t *)keys[0].p, (uint64_t *)keys[1].p, (const uint32_t *)slots[0].p,
                                                  (uint32_t *)slots[1].p, (int)n, 0, endBit, s));
        tmp.ensure(std::max<size_t>(tmpBytes, 16));
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmpBytes, (const uint64_t *)keys[0].p, (uint64_t *)keys[1].p, (const uint32_t *)slots[0].p,
                                                  (uint32_t *)slots[1].p, (int)n, 0, endBit, s));
        const unsigned gridN = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL((k_table_records<C>), dim3(gridN), dim3(256), 0, s, F, (const uint32_t *)slots[1].p, (uint32_t)n, through ? 1 : 0,
                           (const BotRec<C> *)D.genomes[(size_t)dst].bot, (ComposedRec<C> *)out.recs, (C *)out.eo, (unsigned long long *)bad.p);
        hipLaunchKernelGGL((k_table_touch<C>), dim3(gridN), dim3(256), 0, s, (const ComposedRec<C> *)out.recs, (uint32_t)n, shift, out.coarse);
        unsigned long long isBad = 0;
        HIP_OK(hipMemcpyAsync(&isBad, bad.p, 8, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (isBad)
            throw std::runtime_error("internal: a forward source segment produced a source-reversed piece");
    }
    lap("sort + records");
    const unsigned gridB = (unsigned)(((size_t)nb + 1 + 255) / 256);
    hipLaunchKernelGGL((k_table_starts<C>), dim3(gridB), dim3(256), 0, s, (const ComposedRec<C> *)out.recs, (uint32_t)n, shift, nb, out.starts);
    hipLaunchKernelGGL(k_table_fill, dim3(gridB), dim3(256), 0, s, out.coarse, (const uint32_t *)out.starts, nb);
    HIP_OK(hipStreamSynchronize(s));
    if (!through) { // the up table's kernel has no use for starts[]
        (void)hipFree(out.starts);
        out.starts = nullptr;
        bytes -= ((size_t)nb + 1) * 4;
    }
    lap("bucket tables");
    out.shift = shift;
    out.numRecs = n;
    out.through = through;
    h->dev->bytes += bytes;
    out.buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// radix sort of n (key, value) pairs into the second halves of the buffers
static void sortPairs(DevBuf keys[2], DevBuf vals[2], DevBuf &tmp, size_t n, int endBit, hipStream_t s) {
    size_t tmpBytes = 0;
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, (const uint64_t *)keys[0].p, (uint64_t *)keys[1].p, (const uint32_t *)vals[0].p,
                                              (uint32_t *)vals[1].p, (int)n, 0, endBit, s));
    tmp.ensure(std::max<size_t>(tmpBytes, 16));
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmpBytes, (const uint64_t *)keys[0].p, (uint64_t *)keys[1].p, (const uint32_t *)vals[0].p,
                                              (uint32_t *)vals[1].p, (int)n, 0, endBit, s));
}

static int bitsOf(int64_t v) { // smallest b with v < 2^b
    int b = 0;
    while (b < 63 && (v >> b) != 0)
        ++b;
    return b;
}

// The merged form of a whole-path table (hgx_merged_kernels.hpp): chains of mergeable pieces, flags, bucket tables.  All on
// the device: radix sorts over the junction keys and the chains, pointer jumping, a sort by target start and a running maximum
// for the flags.  window: HGX_MERGED_WINDOW (default 8192 bases).  C: the coordinate type of the alignment's tables; the
// 64-bit tables sort their junctions and chains in two stable passes (a junction does not fit one 64-bit key there).
template <typename C> static void buildMerged(hgx_alignment *h, int src, int dst, ComposedUp &c) {
    typedef typename MergeWord<C>::U U;
    constexpr bool WIDE = sizeof(C) == 8;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { g_phases.lap((std::string("merged: ") + what).c_str()); };
    const DeviceImage &D = *h->dev;
    const GenomeTables &S = h->img.genomes[(size_t)src];
    const GenomeTables &T = h->img.genomes[(size_t)dst];
    const DeviceGenome &SG = D.genomes[(size_t)src], &TG = D.genomes[(size_t)dst];
    const size_t n = (size_t)c.numRecs;
    if (n == 0 || n >= ((size_t)1 << 30) || T.seqs.size() >= ((size_t)1 << 24))
        return;
    HIP_OK(hipSetDevice(D.device));
    hipStream_t s = nullptr;
    int64_t window = 8192;
    if (const char *e = getenv("HGX_MERGED_WINDOW"))
        window = std::max<long long>(1, atoll(e));
    const ComposedRec<C> *recs = (const ComposedRec<C> *)c.recs;
    const int sBits = bitsOf(S.totalLength), tBits = bitsOf(T.totalLength);
    DevBuf keys[2], vals[2], tmp, root, minS, sumLen, scalars;
    for (int k = 0; k < 2; ++k) {
        keys[k].ensure(8 * 2 * n);
        vals[k].ensure(4 * 2 * n);
    }
    root.ensure(4 * n);
    minS.ensure(sizeof(U) * n);
    sumLen.ensure(sizeof(U) * n);
    scalars.ensure(16);
    const unsigned gridN = (unsigned)((n + 255) / 256), grid2N = (unsigned)((2 * n + 255) / 256);
    lap("allocations");
    // 1. junctions: which piece continues which
    if constexpr (!WIDE) {
        hipLaunchKernelGGL(k_merge_keys, dim3(gridN), dim3(256), 0, s, recs, (uint32_t)n, sBits, (uint64_t *)keys[0].p, (uint32_t *)vals[0].p);
        sortPairs(keys, vals, tmp, 2 * n, std::min(64, tBits + sBits + 2), s);
    } else {
        hipLaunchKernelGGL(k_merge_keys_low, dim3(gridN), dim3(256), 0, s, recs, (uint32_t)n, (uint64_t *)keys[0].p, (uint32_t *)vals[0].p);
        sortPairs(keys, vals, tmp, 2 * n, std::min(64, sBits + 2), s);
        hipLaunchKernelGGL(k_merge_keys_high, dim3(grid2N), dim3(256), 0, s, recs, (const uint32_t *)vals[1].p, (uint32_t)(2 * n), (uint64_t *)keys[0].p,
                           (uint32_t *)vals[0].p);
        sortPairs(keys, vals, tmp, 2 * n, std::min(64, tBits), s);
    }
    hipLaunchKernelGGL(k_merge_identity, dim3(gridN), dim3(256), 0, s, (uint32_t *)root.p, (uint32_t)n);
    hipLaunchKernelGGL((k_merge_link<C>), dim3(grid2N), dim3(256), 0, s, (const uint32_t *)vals[1].p, (uint32_t)(2 * n), recs,
                       (const int64_t *)TG.seqStart, (int)TG.numSeq, (const int64_t *)SG.seqStart, (int)SG.numSeq, (uint32_t *)root.p);
    lap("junction sort + links");
    // 2. chains: pointer jumping until nothing moves (the flag is read every fourth round)
    for (int round = 0; round < 64; round += 4) {
        HIP_OK(hipMemsetAsync(scalars.p, 0, 4, s));
        for (int k = 0; k < 4; ++k)
            hipLaunchKernelGGL(k_merge_jump, dim3(gridN), dim3(256), 0, s, (uint32_t *)root.p, (uint32_t)n, (unsigned int *)scalars.p);
        unsigned int changed = 0;
        HIP_OK(hipMemcpyAsync(&changed, scalars.p, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (!changed)
            break;
    }
    lap("pointer jumping");
    HIP_OK(hipMemsetAsync(minS.p, 0xFF, sizeof(U) * n, s));
    HIP_OK(hipMemsetAsync(sumLen.p, 0, sizeof(U) * n, s));
    HIP_OK(hipMemsetAsync(scalars.p, 0, 16, s));
    hipLaunchKernelGGL((k_merge_extent<C>), dim3(gridN), dim3(256), 0, s, recs, (const uint32_t *)root.p, (uint32_t)n, (U *)minS.p, (U *)sumLen.p);
    if constexpr (!WIDE) {
        hipLaunchKernelGGL(k_merge_heads, dim3(gridN), dim3(256), 0, s, recs, (const uint32_t *)root.p, (uint32_t)n, (const uint32_t *)minS.p,
                           (uint64_t *)keys[0].p, (uint32_t *)vals[0].p, (unsigned int *)scalars.p);
        sortPairs(keys, vals, tmp, n, 64, s);
    } else {
        hipLaunchKernelGGL(k_merge_heads_low, dim3(gridN), dim3(256), 0, s, recs, (const uint32_t *)root.p, (uint32_t)n, (uint64_t *)keys[0].p,
                           (uint32_t *)vals[0].p, (unsigned int *)scalars.p);
        sortPairs(keys, vals, tmp, n, 64, s);
        hipLaunchKernelGGL(k_merge_heads_high, dim3(gridN), dim3(256), 0, s, (const uint32_t *)root.p, (const uint32_t *)vals[1].p, (uint32_t)n,
                           (const unsigned long long *)minS.p, (uint64_t *)keys[0].p, (uint32_t *)vals[0].p);
        sortPairs(keys, vals, tmp, n, 64, s);
    }
    unsigned int m32 = 0;
    HIP_OK(hipMemcpyAsync(&m32, scalars.p, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    const size_t m = m32;
    HIP_OK(hipMalloc(&c.mRecs, (m + LIFT_SENTINELS) * sizeof(ComposedRec<C>)));
    ComposedRec<C> *mrecs = (ComposedRec<C> *)c.mRecs;
    const unsigned gridM = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL((k_merge_records<C>), dim3(gridM), dim3(256), 0, s, recs, (const uint32_t *)vals[1].p, (uint32_t)m, (const U *)minS.p,
                       (const U *)sumLen.p, (const int64_t *)TG.seqStart, (int)TG.numSeq, mrecs);
    lap("chains sorted, records");
    // 3. flags: records whose target range overlaps that of a record nearby in the source
    DevBuf flag, flagPrefix, tHi, runMax;
    flag.ensure(4 * (m + 1));
    flagPrefix.ensure(4 * (m + 1));
    tHi.ensure(sizeof(U) * std::max<size_t>(m, 1));
    runMax.ensure(sizeof(U) * std::max<size_t>(m, 1));
    HIP_OK(hipMemsetAsync(flag.p, 0, 4 * (m + 1), s));
    hipLaunchKernelGGL((k_flag_keys<C>), dim3(gridM), dim3(256), 0, s, (const ComposedRec<C> *)mrecs, (uint32_t)m, (uint64_t *)keys[0].p, (uint32_t *)vals[0].p);
    sortPairs(keys, vals, tmp, m, std::min(64, std::max(1, tBits)), s);
    hipLaunchKernelGGL((k_flag_ends<C>), dim3(gridM), dim3(256), 0, s, (const ComposedRec<C> *)mrecs, (const uint32_t *)vals[1].p, (uint32_t)m, (U *)tHi.p);
    {
        size_t tmpBytes = 0;
        HIP_OK(hipcub::DeviceScan::InclusiveScan(nullptr, tmpBytes, (const U *)tHi.p, (U *)runMax.p, MaxOp(), (int)m, s));
        tmp.ensure(std::max<size_t>(tmpBytes, 16));
        HIP_OK(hipcub::DeviceScan::InclusiveScan(tmp.p, tmpBytes, (const U *)tHi.p, (U *)runMax.p, MaxOp(), (int)m, s));
    }
    hipLaunchKernelGGL((k_flag_overlaps<C>), dim3(gridM), dim3(256), 0, s, (const ComposedRec<C> *)mrecs, (const uint64_t *)keys[1].p,
                       (const uint32_t *)vals[1].p, (const U *)runMax.p, (uint32_t)m, window, (uint32_t *)flag.p);
    {
        size_t tmpBytes = 0;
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, (const uint32_t *)flag.p, (uint32_t *)flagPrefix.p, (int)(m + 1), s));
        tmp.ensure(std::max<size_t>(tmpBytes, 16));
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmpBytes, (const uint32_t *)flag.p, (uint32_t *)flagPrefix.p, (int)(m + 1), s));
    }
    lap("flags");
    // 4. bucket tables, about one record per bucket (an interval's reach is then a record or two wider than its records)
    int64_t perBucket = 1;
    if (const char *e = getenv("HGX_MERGED_BUCKET_RECS")) // records per bucket (experiments)
        perBucket = std::max<long long>(1, atoll(e));
    int64_t buckets = 1;
    while (buckets < (int64_t)m / perBucket && buckets < ((int64_t)1 << 22))
        buckets <<= 1;
    int shift = 0;
    while (((S.totalLength - 1) >> shift) >= buckets)
        ++shift;
    const uint32_t nb = (uint32_t)(((S.totalLength - 1) >> shift) + 1);
    DevBuf coarse, starts;
    coarse.ensure(((size_t)nb + 1) * 4);
    starts.ensure(((size_t)nb + 1) * 4);
    HIP_OK(hipMemsetAsync(coarse.p, 0xFF, ((size_t)nb + 1) * 4, s));
    HIP_OK(hipMalloc(&c.mBuckets, ((size_t)nb + 1) * 4));
    const size_t flagBitsBytes = 4 * (((size_t)nb + 1) / 32 + 3);
    HIP_OK(hipMalloc(&c.mFlagBits, flagBitsBytes));
    HIP_OK(hipMemsetAsync(c.mFlagBits, 0, flagBitsBytes, s));
    if (m)
        hipLaunchKernelGGL((k_table_touch<C>), dim3(gridM), dim3(256), 0, s, (const ComposedRec<C> *)mrecs, (uint32_t)m, shift, (uint32_t *)coarse.p);
    const unsigned gridB = (unsigned)(((size_t)nb + 1 + 255) / 256);
    hipLaunchKernelGGL((k_table_starts<C>), dim3(gridB), dim3(256), 0, s, (const ComposedRec<C> *)mrecs, (uint32_t)m, shift, nb, (uint32_t *)starts.p);
    hipLaunchKernelGGL(k_table_fill, dim3(gridB), dim3(256), 0, s, (uint32_t *)coarse.p, (const uint32_t *)starts.p, nb);
    HIP_OK(hipMemcpyAsync(c.mBuckets, coarse.p, ((size_t)nb + 1) * 4, hipMemcpyDeviceToDevice, s));
    if (m)
        hipLaunchKernelGGL((k_bucket_flag_bits<C>), dim3(gridM), dim3(256), 0, s, (const ComposedRec<C> *)mrecs, (const uint32_t *)flag.p, (uint32_t)m, shift,
                           (uint32_t *)c.mFlagBits);
    hipLaunchKernelGGL((k_merge_mark<C>), dim3((unsigned)((m + LIFT_SENTINELS + 255) / 256)), dim3(256), 0, s, mrecs, (const uint32_t *)flag.p, (uint32_t)m);
    unsigned int flagged = 0;
    HIP_OK(hipMemcpyAsync(&flagged, (const uint32_t *)flagPrefix.p + m, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    lap("bucket tables");
    c.mShift = shift;
    c.mNum = m;
    c.mFlagged = flagged;
    c.mWindow = window;
    h->dev->bytes += (m + LIFT_SENTINELS) * sizeof(ComposedRec<C>) + ((size_t)nb + 1) * 4 + flagBitsBytes;
    c.mBuildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static const ComposedUp *ensureComposed(hgx_alignment *h, int src, int dst, bool through, const hgx_liftover_opts &opts, bool wantMerged) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    // the merged form for the single-pass kernels (HGX_MERGED=0: keep to the multi-kernel path), built when a plan asks for it
    auto addMerged = [&](ComposedUp &c) {
        const char *me = getenv("HGX_MERGED");
        if (wantMerged && through && !c.mRecs && !c.mTried && !(me && me[0] == '0')) {
            c.mTried = true;
            if (h->dev->wide)
                buildMerged<int64_t>(h, src, dst, c);
            else
                buildMerged<int32_t>(h, src, dst, c);
        }
    };
    const bool climbs = opts.coalescence_limit >= 0 && opts.traverse_dupes; // createLiftoverPlan ignores the limit without dupes
    const std::array<int, 4> key = through ? std::array<int, 4>{src, dst, (opts.traverse_dupes ? 1 : 0) | (opts.block_mapper_source ? 2 : 0),
                                                                  climbs ? opts.coalescence_limit + 1 : 0}
                                           : std::array<int, 4>{src, dst, -1, -1};
    auto it = h->dev->composed.find(key);
    if (it != h->dev->composed.end()) {
        addMerged(it->second);
        return &it->second;
    }
    ComposedUp c;
    if (h->dev->wide)
        buildComposed<int64_t>(h, src, dst, through, opts, c);
    else
        buildComposed<int32_t>(h, src, dst, through, opts, c);
    g_phases.lap("pieces: builder plan and workspaces released");
    addMerged(c);
    return &h->dev->composed.emplace(key, c).first->second;
}

// The table of the whole path src -> dst (dupes on, paralogs followed up to `limit`, -1 = the MRCA) for users outside the
// liftover plans (the depth of hgx_columns.hip); null when this pair cannot have one (the source is the MRCA or has no top
// tiling — the conditions of createLiftoverPlan).
const ComposedUp *wholePathTable(hgx_alignment *h, int src, int dst, int limit) {
    const Image &img = h->img;
    if (src == dst || img.lca(src, dst) == src || img.genomes[(size_t)src].numTop <= 0)
        return nullptr;
    hgx_liftover_opts o{};
    o.traverse_dupes = 1;
    o.coalescence_limit = (limit >= 0 && limit != img.lca(src, dst)) ? limit : -1;
    // an ancestor of the source is reached by the up walk alone: the up table (its records are the pieces that arrive there;
    // paralogs found higher up cannot add an image where the direct lineage has none)
    if (img.lca(src, dst) == dst)
        return ensureComposed(h, src, dst, false, o, false);
    return ensureComposed(h, src, dst, true, o, /*wantMerged=*/false);
}

// hgx_liftover_submit: queues a batch's launches on `stream` and returns; hgx_liftover_collect waits for them.  A plan has one
// batch in flight; two plans on two streams keep the GPU busy across the ends of their launches (the counting launch ends with
// a few wavefronts finishing general intervals, the host's launch and wake-up times sit between batches).  Only the steady
// state of the single-pass path is queued — anything else (no table yet, kernel events on, a run that needs one of the
// repeat-with-larger-buffers paths) is run to the end by submit, or repeated to the end by collect.
void submitLiftoverPlan(hgx_liftover_plan *p, size_t n, const int64_t *dS, const int64_t *dE, const uint8_t *dStrand, void *stream) {
    hgx_liftover_plan &P = *p;
    if (P.pendingState != 0)
        throw std::runtime_error("hgx_liftover_submit: the plan's previous batch has not been collected");
    P.pendingN = n;
    P.pendingS = dS;
    P.pendingE = dE;
    P.pendingStrand = dStrand;
    P.pendingStream = (hipStream_t)stream;
    const bool steady = n > 0 && n <= P.maxQueries && n < ((size_t)1 << 31) && P.composed && P.composed->mRecs && !P.opts.emit_blocks &&
                        P.timer.mode == 0 && !P.captureUp && !P.captureFinal;
    if (!steady) {
        runLiftoverPlan(p, n, dS, dE, dStrand, stream, &P.pendingOut, &P.pendingCount);
        P.pendingState = 2;
        return;
    }
    HIP_OK(hipSetDevice(P.h->dev->device));
    std::vector<unsigned long long> hc(CNT_SLOTS);
    P.mergedOffThisRun = false;
    P.pendingState = 1; // (from the first launch on the plan's workspaces belong to this batch)
    try {
        if (P.h->dev->wide)
            runMergedOnce<int64_t>(P, n, dS, dE, dStrand, (hipStream_t)stream, hc.data(), true);
        else
            runMergedOnce<int32_t>(P, n, dS, dE, dStrand, (hipStream_t)stream, hc.data(), true);
    } catch (...) {
        // some launches may be queued: let them finish before anybody touches the workspaces again, and forget the batch
        (void)hipStreamSynchronize((hipStream_t)stream);
        P.pendingState = 0;
        P.liftStateClean = false;
        throw;
    }
}

void collectLiftoverPlan(hgx_liftover_plan *p, const hgx_record **dOut, size_t *nOut) {
    hgx_liftover_plan &P = *p;
    if (P.pendingState == 0)
        throw std::runtime_error("hgx_liftover_collect: nothing was submitted");
    const int state = P.pendingState;
    P.pendingState = 0;
    HIP_OK(hipSetDevice(P.h->dev->device)); // (the caller's thread may be bound to another device; the rerun path launches)
    if (state == 2) {
        *dOut = P.pendingOut;
        *nOut = P.pendingCount;
        return;
    }
    std::vector<unsigned long long> hc(CNT_SLOTS);
    finishMergedOnce(P, P.pendingStream, hc.data());
    // (runPlan's conditions for repeating a batch: buffers too small, the scratch of k_finish_big too small, intervals passed on
    // to launches that were not made — rare: the batch again, through every repeat path runPlan has)
    if (hc[CNT_OVERFLOW] || (hc[CNT_DEFERRED] && (hc[CNT_BIGFAIL] || hc[CNT_DEFERRED] > P.liftBigSlots)) || (P.liftRestCount && P.liftRestSkipped)) {
        runLiftoverPlan(p, P.pendingN, P.pendingS, P.pendingE, P.pendingStrand, P.pendingStream, dOut, nOut);
        return;
    }
    P.stats = hgx_liftover_stats{};
    P.stats.queries = P.pendingN;
    P.stats.mapped_pieces = hc[CNT_MAPPED];
    for (int k = 0; k < MAX_LAUNCHES; ++k) {
        P.stats.top_derefs += hc[CNT_KSTAT0 + 2 * k];
        P.stats.bottom_derefs += hc[CNT_KSTAT0 + 2 * k + 1];
    }
    P.stats.records = (uint32_t)(P.pinned[CNT_SLOTS] & 0xFFFFFFFFull);
    P.stats.deferred_queries = (uint32_t)hc[CNT_DEFERRED];
    if (P.liftRestCount)
        P.liftRestSeen = true;
    P.stats.composed_kind = 3;
    P.stats.composed_records = P.composed->mNum;
    P.stats.composed_build_ms = P.composed->buildMs + P.composed->mBuildMs;
    P.stats.general_queries = P.generalQueries;
    P.stats.composed_flagged = P.composed->mFlagged;
    *dOut = (const hgx_record *)P.outRecords.p;
    *nOut = P.stats.records;
}

void runLiftoverPlan(hgx_liftover_plan *p, size_t n, const int64_t *dS, const int64_t *dE, const uint8_t *dStrand, void *stream,
                     const hgx_record **dOut, size_t *nOut) {
    if (p->pendingState == 1) // (its workspaces and its output belong to the batch in flight)
        throw std::runtime_error("the plan has a submitted batch that has not been collected");
    p->pendingState = 0; // (a batch submit ran to the end and nobody collected is superseded by this run: collect then has nothing)
    if (p->h->dev->wide)
        runPlan<int64_t>(*p, n, dS, dE, dStrand, (hipStream_t)stream, dOut, nOut);
    else
        runPlan<int32_t>(*p, n, dS, dE, dStrand, (hipStream_t)stream, dOut, nOut);
}

void destroyLiftoverPlan(hgx_liftover_plan *p) {
    if (!p)
        return;
    (void)hipSetDevice(p->device);
    if (p->pendingState == 1) // (a submitted batch nobody collected: its launches read and write the plan's workspaces)
        (void)hipStreamSynchronize(p->pendingStream);
    delete p;
}

const hgx_liftover_stats &liftoverPlanStats(const hgx_liftover_plan *p) {
    return p->stats;
}

static void refuseWhileInFlight(const hgx_liftover_plan *p, const char *who) {
    // (stats and outRecords belong to the batch in flight: its launches are still writing the records and k_lift_totals has not
    // reported yet)
    if (p->pendingState == 1)
        throw std::runtime_error(std::string(who) + ": the plan has a submitted batch that has not been collected");
}

void liftoverPlanCopyRecords(const hgx_liftover_plan *p, void *dDst, size_t nRecords, void *stream) {
    refuseWhileInFlight(p, "hgx_liftover_copy_records");
    if (nRecords > p->stats.records)
        throw std::runtime_error("more records requested than the last run produced");
    HIP_OK(hipSetDevice(p->device));
    if (nRecords)
        HIP_OK(hipMemcpyAsync(dDst, p->outRecords.p, nRecords * sizeof(hgx_record), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
}

void liftoverPlanSetWorkers(hgx_liftover_plan *p, int n) {
    p->liftWorkersWanted = n;
}

void liftoverPlanSetTiming(hgx_liftover_plan *p, int mode) {
    if (mode < 0 || mode > 2)
        throw std::runtime_error("timing mode must be 0 (off), 1 (last run) or 2 (accumulate)");
    HIP_OK(hipSetDevice(p->device));
    p->timer.read(); // closes whatever was pending
    p->timer.recs.clear();
    p->timer.used = 0;
    p->timer.runStart = 0;
    p->timer.mode = mode;
}

void liftoverPlanCopyRecordsPacked(const hgx_liftover_plan *p, void *dDst, size_t nRecords, void *stream) {
    refuseWhileInFlight(p, "hgx_liftover_copy_records_packed");
    if (nRecords > p->stats.records)
        throw std::runtime_error("more records requested than the last run produced");
    HIP_OK(hipSetDevice(p->device));
    if (nRecords)
        hipLaunchKernelGGL(k_pack_records, dim3(GRID), dim3(256), 0, (hipStream_t)stream, (const hgx_record *)p->outRecords.p, (uint32_t)nRecords,
                           (int32_t *)dDst);
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
}

// The last run's records as one self-describing blob for the multi-GPU exchange (include/hgx.h: hgx_liftover_wire_blob).
size_t liftoverPlanWireBlob(hgx_liftover_plan *p, void *dDst, size_t capacity, int64_t firstQuery, int *format, void *stream) {
    refuseWhileInFlight(p, "hgx_liftover_wire_blob");
    HIP_OK(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t nrec = (size_t)p->stats.records, nq = (size_t)p->stats.queries;
    const size_t countsBytes = (2 * nq + 7) & ~(size_t)7;
    const size_t need12 = 32 + countsBytes + 12 * nrec, need20 = 32 + 20 * nrec, need40 = 32 + sizeof(hgx_record) * nrec;
    // the 20-byte form holds 31-bit coordinates and a 16-bit sequence index; beyond that the records travel as they are
    const Image &img = p->h->img;
    const bool fits20 = img.genomes[(size_t)p->src].totalLength < ((int64_t)1 << 31) && img.genomes[(size_t)p->tgt].totalLength < ((int64_t)1 << 31) &&
                        img.genomes[(size_t)p->tgt].seqs.size() <= ((size_t)1 << 16);
    const char *forced = getenv("HGX_WIRE_FORMAT"); // tests: the wider forms on data that would fit the narrow one
    const size_t needFallback = (fits20 && !(forced && forced[0] == '4')) ? need20 : need40;
    if (!dDst)
        return std::max(need12, needFallback); // capacity query
    if (capacity < std::max(need12, needFallback))
        throw std::runtime_error("hgx_liftover_wire_blob: destination too small");
    unsigned char *dst = (unsigned char *)dDst;
    int fmt = 12;
    const bool bedOnly = format && *format == 8; // the caller asks for the 8-byte form (no source coordinates: BED lines only)
    {
        p->wireFlag.ensure(4);
        if (!p->wireFlagHost)
            HIP_OK(hipHostMalloc((void **)&p->wireFlagHost, 4));
        HIP_OK(hipMemsetAsync(p->wireFlag.p, 0, 4, s));
        if (nq)
            hipLaunchKernelGGL(k_wire12_counts, dim3(GRID), dim3(256), 0, s, (const uint32_t *)p->nOut.p, (uint32_t)nq, (uint32_t)(countsBytes / 2),
                               (uint16_t *)(dst + 32), (unsigned int *)p->wireFlag.p);
        if (nrec && bedOnly)
            hipLaunchKernelGGL(k_wire8_records, dim3(GRID), dim3(256), 0, s, (const hgx_record *)p->outRecords.p, (uint32_t)nrec,
                               (uint32_t *)(dst + 32 + countsBytes), (unsigned int *)p->wireFlag.p);
        else if (nrec)
            hipLaunchKernelGGL(k_wire12_records, dim3(GRID), dim3(256), 0, s, (const hgx_record *)p->outRecords.p, (uint32_t)nrec,
                               (uint32_t *)(dst + 32 + countsBytes), (unsigned int *)p->wireFlag.p);
        HIP_OK(hipMemcpyAsync(p->wireFlagHost, p->wireFlag.p, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s)); // the one synchronisation of the call: does the narrow form hold this batch?
        if (*p->wireFlagHost)
            fmt = fits20 ? 20 : 40;
        else if (bedOnly)
            fmt = 8;
    }
    if (forced && forced[0] == '4')
        fmt = 40;
    else if (forced && forced[0] == '2' && fits20)
        fmt = 20;
    if (fmt == 20 && nrec)
        hipLaunchKernelGGL(k_pack_records, dim3(GRID), dim3(256), 0, s, (const hgx_record *)p->outRecords.p, (uint32_t)nrec, (int32_t *)(dst + 32));
    if (fmt == 40 && nrec)
        HIP_OK(hipMemcpyAsync(dst + 32, p->outRecords.p, sizeof(hgx_record) * nrec, hipMemcpyDeviceToDevice, s));
    const WireHeader header = {{'H', 'G', 'X', 'W'}, (uint32_t)fmt, firstQuery, (uint64_t)nq, (uint64_t)nrec};
    hipLaunchKernelGGL(k_wire_header, dim3(1), dim3(1), 0, s, header, (WireHeader *)dst); // (stream ordered: no further wait)
    if (format)
        *format = fmt;
    return fmt == 8 ? 32 + countsBytes + 8 * nrec : fmt == 12 ? need12 : fmt == 20 ? need20 : need40;
}

std::string liftoverPlanKernelTimes(hgx_liftover_plan *p) {
    HIP_OK(hipSetDevice(p->device));
    std::string s = "{";
    bool first = true;
    for (auto &kv : p->timer.read()) {
        if (!first)
            s += ", ";
        first = false;
        char buf[256];
        snprintf(buf, sizeof buf, "\"%s\": {\"ms\": %.6f, \"launches\": %d, \"top_derefs\": %llu, \"bot_derefs\": %llu}",
                 kv.first.c_str(), kv.second.ms, kv.second.launches, kv.second.top, kv.second.bot);
        s += buf;
    }
    s += "}";
    return s;
}

// host-buffer batch: H2D, run, D2H
// alloc(n) returns the host memory the n records are copied into (straight from the device: no staging copy)
static void runHostArraysInto(hgx_liftover_plan *P, const std::vector<int64_t> &gs, const std::vector<int64_t> &ge, const std::vector<uint8_t> &st,
                              const std::function<hgx_record *(size_t)> &alloc) {
    const size_t n = gs.size();
    DevBuf dS, dE, dT;
    dS.ensure(8 * std::max<size_t>(n, 1));
    dE.ensure(8 * std::max<size_t>(n, 1));
    dT.ensure(std::max<size_t>(n, 1));
    HIP_OK(hipMemcpy(dS.p, gs.data(), 8 * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dE.p, ge.data(), 8 * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dT.p, st.data(), n, hipMemcpyHostToDevice));
    const hgx_record *dOut = nullptr;
    size_t nOut = 0;
    runLiftoverPlan(P, n, (const int64_t *)dS.p, (const int64_t *)dE.p, (const uint8_t *)dT.p, nullptr, &dOut, &nOut);
    hgx_record *dst = alloc(nOut);
    if (nOut)
        HIP_OK(hipMemcpy(dst, dOut, sizeof(hgx_record) * nOut, hipMemcpyDeviceToHost));
}

static void runHostArrays(hgx_liftover_plan *P, const std::vector<int64_t> &gs, const std::vector<int64_t> &ge, const std::vector<uint8_t> &st,
                          std::vector<hgx_record> &out) {
    const size_t n = gs.size();
    DevBuf dS, dE, dT;
    dS.ensure(8 * std::max<size_t>(n, 1));
    dE.ensure(8 * std::max<size_t>(n, 1));
    dT.ensure(std::max<size_t>(n, 1));
    HIP_OK(hipMemcpy(dS.p, gs.data(), 8 * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dE.p, ge.data(), 8 * n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dT.p, st.data(), n, hipMemcpyHostToDevice));
    const hgx_record *dOut = nullptr;
    size_t nOut = 0;
    runLiftoverPlan(P, n, (const int64_t *)dS.p, (const int64_t *)dE.p, (const uint8_t *)dT.p, nullptr, &dOut, &nOut);
    out.resize(nOut);
    if (nOut)
        HIP_OK(hipMemcpy(out.data(), dOut, sizeof(hgx_record) * nOut, hipMemcpyDeviceToHost));
}

} // namespace hgx

hgx_alignment::~hgx_alignment() {
    mafTracks.reset(); // (device workspaces: back to the block cache before it is trimmed)
    if (stage.packed)
        (void)hipHostFree(stage.packed);
    if (stage.first)
        (void)hipHostFree(stage.first);
    if (stage.flagHost)
        (void)hipHostFree(stage.flagHost);
    if (stage.dPacked)
        (void)hipFree(stage.dPacked);
    if (stage.dFlag)
        (void)hipFree(stage.dFlag);
    if (cachedPlan.plan)
        hgx::destroyLiftoverPlan(cachedPlan.plan);
    for (CachedPlan &c : vizPlans)
        if (c.plan)
            hgx::destroyLiftoverPlan(c.plan);
    if (stage.gs)
        (void)hipHostFree(stage.gs);
    if (stage.ge)
        (void)hipHostFree(stage.ge);
    if (stage.st)
        (void)hipHostFree(stage.st);
    if (stage.recs)
        (void)hipHostFree(stage.recs);
    if (stage.dS)
        (void)hipFree(stage.dS);
    if (stage.dE)
        (void)hipFree(stage.dE);
    if (stage.dT)
        (void)hipFree(stage.dT);
}

namespace hgx {

// sequence-relative half-open intervals -> inclusive genome coordinates; invalid ones become empty (gs = 0, ge = -1)
static void intervalsToGenomeCoordinates(const GenomeTables &G, size_t n, const hgx_interval *iv, std::vector<int64_t> &gs,
                                         std::vector<int64_t> &ge, std::vector<uint8_t> &st) {
    gs.resize(n);
    ge.resize(n);
    st.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const hgx_interval &q = iv[i];
        // (a negative start is not checked by the reference: Liftover::visitLine, halLiftover.cpp:52-66, only looks at the end, and
        // liftInterval adds the sequence's start (halBlockLiftover.cpp:48) — the interval then begins in the sequence in front.
        // Same arithmetic here and in the text path; only a start in front of the genome's first base, where the reference's
        // toSite has nothing to stand on, makes the interval empty)
        bool ok = q.seq >= 0 && q.seq < (int32_t)G.seqs.size() && q.start < q.end;
        if (ok)
            ok = q.end <= G.seqs[(size_t)q.seq].length; // halLiftover.cpp:62-66: skipped, not an error
        if (ok)
            ok = q.start + G.seqs[(size_t)q.seq].start >= 0;
        if (ok) {
            gs[i] = q.start + G.seqs[(size_t)q.seq].start;       // halBlockLiftover.cpp:48
            ge[i] = q.end - 1 + G.seqs[(size_t)q.seq].start;     // :49
        } else {
            gs[i] = 0;
            ge[i] = -1;
        }
        st[i] = (uint8_t)q.strand;
    }
}

// the alignment's cached plan for (src, tgt, opts), recreated when the key changes or the batch outgrows it; the
// caller holds h->planMutex while it uses the plan
static hgx_liftover_plan *cachedPlanFor(hgx_alignment *h, int src, int tgt, const hgx_liftover_opts &opts, size_t n, int vizSlot = -1) {
    hgx_alignment::CachedPlan &c = vizSlot < 0 ? h->cachedPlan : h->vizPlans[vizSlot];
    const bool same = c.plan && c.src == src && c.tgt == tgt && memcmp(&c.opts, &opts, sizeof opts) == 0 && n <= c.maxQueries;
    if (!same) {
        if (c.plan)
            destroyLiftoverPlan(c.plan);
        c.plan = nullptr;
        c.plan = createLiftoverPlan(h, src, tgt, opts, n);
        c.src = src;
        c.tgt = tgt;
        c.opts = opts;
        c.maxQueries = std::max<size_t>(n, 1);
    }
    return c.plan;
}

void liftoverBatchHost(hgx_alignment *h, int src, int tgt, size_t n, const hgx_interval *iv, const hgx_liftover_opts &opts,
                       std::vector<hgx_record> &out, hgx_liftover_stats *stats) {
    std::lock_guard<std::mutex> lock(h->planMutex);
    hgx_liftover_plan *P = cachedPlanFor(h, src, tgt, opts, n);
    std::vector<int64_t> gs, ge;
    std::vector<uint8_t> st;
    intervalsToGenomeCoordinates(h->img.genomes[(size_t)src], n, iv, gs, ge, st);
    runHostArrays(P, gs, ge, st, out);
    if (stats)
        *stats = P->stats;
}

void liftoverBatchHostRaw(hgx_alignment *h, int src, int tgt, size_t n, const hgx_interval *iv, const hgx_liftover_opts &opts,
                          const std::function<hgx_record *(size_t)> &alloc) {
    std::lock_guard<std::mutex> lock(h->planMutex);
    hgx_liftover_plan *P = cachedPlanFor(h, src, tgt, opts, n);
    std::vector<int64_t> gs, ge;
    std::vector<uint8_t> st;
    intervalsToGenomeCoordinates(h->img.genomes[(size_t)src], n, iv, gs, ge, st);
    runHostArraysInto(P, gs, ge, st, alloc);
}

void liftoverStageQueries(hgx_alignment *h, size_t n, int64_t **gs, int64_t **ge, uint8_t **strand) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); liftover needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    hgx_alignment::Stage &S = h->stage;
    if (n > S.capQ) {
        const size_t cap = std::max<size_t>(n + n / 4, 1u << 16);
        if (S.gs)
            (void)hipHostFree(S.gs);
        if (S.ge)
            (void)hipHostFree(S.ge);
        if (S.st)
            (void)hipHostFree(S.st);
        S.gs = S.ge = nullptr;
        S.st = nullptr;
        S.capQ = 0;
        HIP_OK(hipHostMalloc((void **)&S.gs, 8 * cap));
        HIP_OK(hipHostMalloc((void **)&S.ge, 8 * cap));
        HIP_OK(hipHostMalloc((void **)&S.st, cap));
        S.capQ = cap;
    }
    *gs = S.gs;
    *ge = S.ge;
    *strand = S.st;
}

void liftoverBatchStaged(hgx_alignment *h, int src, int tgt, size_t n, const hgx_liftover_opts &opts, const hgx_record **recs, size_t *nRecs,
                         hgx_liftover_stats *stats, PackedRecords *packed) {
    std::lock_guard<std::mutex> lock(h->planMutex);
    hgx_alignment::Stage &S = h->stage;
    if (n > S.capQ)
        throw std::runtime_error("liftoverBatchStaged: more intervals than were staged");
    hgx_liftover_plan *P = cachedPlanFor(h, src, tgt, opts, n);
    HIP_OK(hipSetDevice(h->dev->device));
    if (n > S.capD) {
        if (S.dS)
            (void)hipFree(S.dS);
        if (S.dE)
            (void)hipFree(S.dE);
        if (S.dT)
            (void)hipFree(S.dT);
        S.dS = S.dE = S.dT = nullptr;
        S.capD = 0;
        HIP_OK(hipMalloc(&S.dS, 8 * S.capQ));
        HIP_OK(hipMalloc(&S.dE, 8 * S.capQ));
        HIP_OK(hipMalloc(&S.dT, S.capQ));
        S.capD = S.capQ;
    }
    hipStream_t s = nullptr;
    HIP_OK(hipMemcpyAsync(S.dS, S.gs, 8 * n, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemcpyAsync(S.dE, S.ge, 8 * n, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemcpyAsync(S.dT, S.st, n, hipMemcpyHostToDevice, s));
    const hgx_record *dOut = nullptr;
    size_t nOut = 0;
    runLiftoverPlan(P, n, (const int64_t *)S.dS, (const int64_t *)S.dE, (const uint8_t *)S.dT, s, &dOut, &nOut);
    if (packed)
        *packed = PackedRecords{};
    // A caller that prints BED lines: the records leave the device in the 8-byte form of the wire when they fit it (the device
    // checks) and the plan wrote them densely with every interval's first record (the single-pass path): a fifth of the bytes
    // that cross PCIe, which is what this call waits for.
    if (packed && nOut && P->stats.composed_kind == 3 && !getenv("HGX_TEXT_FULL_RECORDS")) {
        if (nOut > S.capP) {
            if (S.packed)
                (void)hipHostFree(S.packed);
            if (S.dPacked)
                (void)hipFree(S.dPacked);
            S.packed = nullptr;
            S.dPacked = nullptr;
            S.capP = 0;
            const size_t cap = nOut + nOut / 4;
            HIP_OK(hipHostMalloc((void **)&S.packed, 8 * cap));
            HIP_OK(hipMalloc(&S.dPacked, 8 * cap));
            S.capP = cap;
        }
        if (n + 1 > S.capF) {
            if (S.first)
                (void)hipHostFree(S.first);
            S.first = nullptr;
            S.capF = 0;
            HIP_OK(hipHostMalloc((void **)&S.first, 4 * (S.capQ + 1)));
            S.capF = S.capQ + 1;
        }
        if (!S.dFlag) {
            HIP_OK(hipMalloc(&S.dFlag, 8));
            HIP_OK(hipHostMalloc((void **)&S.flagHost, 8));
        }
        HIP_OK(hipMemsetAsync(S.dFlag, 0, 4, s));
        hipLaunchKernelGGL(k_wire8_records, dim3(GRID), dim3(256), 0, s, dOut, (uint32_t)nOut, (uint32_t *)S.dPacked, (unsigned int *)S.dFlag);
        HIP_OK(hipMemcpyAsync(S.flagHost, S.dFlag, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(S.first, P->outOffset.p, 4 * n, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(S.packed, S.dPacked, 8 * nOut, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (!*S.flagHost) {
            S.first[n] = (uint32_t)nOut;
            packed->words = S.packed;
            packed->first = S.first;
            *recs = nullptr;
            *nRecs = nOut;
            if (stats)
                *stats = P->stats;
            return;
        }
        // (a field does not fit — 128 target sequences or more, a record of 4 M bases: the records as they are, below)
    }
    if (nOut > S.capR) {
        if (S.recs)
            (void)hipHostFree(S.recs);
        S.recs = nullptr;
        S.capR = 0;
        const size_t cap = nOut + nOut / 4;
        HIP_OK(hipHostMalloc((void **)&S.recs, sizeof(hgx_record) * cap));
        S.capR = cap;
    }
    if (nOut) {
        HIP_OK(hipMemcpyAsync(S.recs, dOut, sizeof(hgx_record) * nOut, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
    }
    *recs = S.recs;
    *nRecs = nOut;
    if (stats)
        *stats = P->stats;
}

// a batch of absolute intervals (inclusive genome coordinates of src) through a plan of its own, records to the host
void liftoverBatchAbsolute(hgx_alignment *h, int src, int tgt, const std::vector<int64_t> &gs, const std::vector<int64_t> &ge,
                           const std::vector<uint8_t> &strand, const hgx_liftover_opts &opts, std::vector<hgx_record> &out) {
    out.clear();
    if (gs.empty())
        return;
    // (the two plans of halGetBlocksInTargetRange — emit_blocks 1: the range forward, 2: the neighbours back — are kept with the
    // alignment; the calls are serialised as the reference's are, blockViz/impl/halBlockViz.cpp: halLock)
    std::lock_guard<std::mutex> lock(h->planMutex);
    hgx_liftover_plan *P = cachedPlanFor(h, src, tgt, opts, std::max<size_t>(gs.size(), 64), opts.emit_blocks == 2 ? 1 : 0);
    runHostArrays(P, gs, ge, strand, out);
}

void blockMapHost(hgx_alignment *h, int ref, int query, int64_t absFirst, int64_t absLast, bool targetReversed,
                  const hgx_liftover_opts &optsIn, std::vector<hgx_record> &out) {
    hgx_liftover_opts opts = optsIn;
    opts.emit_blocks = 1;
    opts.block_mapper_source = 1;
    const GenomeTables &G = h->img.genomes[(size_t)ref];
    if (absFirst < 0 || absLast < absFirst || absLast >= G.totalLength)
        throw std::runtime_error("hgx_block_map: reference range out of bounds");
    std::unique_ptr<hgx_liftover_plan, void (*)(hgx_liftover_plan *)> P(createLiftoverPlan(h, ref, query, opts, 1), destroyLiftoverPlan);
    std::vector<int64_t> gs{absFirst}, ge{absLast};
    std::vector<uint8_t> st{(uint8_t)(targetReversed ? '-' : '+')};
    runHostArrays(P.get(), gs, ge, st, out);
}

} // namespace hgx
