// Column engine: launches of the per-base closure kernels behind hgx_columns_depth / hgx_alignment_depth /
// hgx_maf_export (include/hgx.h).
#include "hgx_column_kernels.hpp"
#include "hgx_columns_engine.hpp"
#include <algorithm>
#include <cstring>
#include <functional>
#include <set>

namespace hgx {

#define HIP_OK(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);               \
    } while (0)

namespace {
struct Buf {
    void *p = nullptr;
    explicit Buf(size_t bytes) {
        HIP_OK(hipMalloc(&p, std::max<size_t>(bytes, 16)));
    }
    ~Buf() {
        if (p)
            (void)hipFree(p);
    }
    Buf(const Buf &) = delete;
};
struct Ev {
    hipEvent_t e;
    Ev() {
        HIP_OK(hipEventCreate(&e));
    }
    ~Ev() {
        (void)hipEventDestroy(e);
    }
};
} // namespace

static ColumnParams makeParams(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, const ColumnOptions &opt,
                               unsigned int *dErr) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    const Image &img = h->img;
    const int ng = (int)img.genomes.size();
    if (ref < 0 || ref >= ng)
        throw std::runtime_error("reference genome id out of range");
    if (ng > 256)
        throw std::runtime_error("alignments with more than 256 genomes are not supported by the column kernels yet");
    const GenomeTables &R = img.genomes[(size_t)ref];
    if (count < 0 || step < 1 || first < 0 || (count > 0 && first + (count - 1) * step >= R.totalLength))
        throw std::runtime_error("column range out of bounds for genome " + R.name);
    ColumnParams P;
    memset(&P, 0, sizeof P);
    P.desc = h->dev->desc;
    P.numGenomes = ng;
    P.ref = ref;
    P.first = first;
    P.count = count;
    P.step = step;
    P.noDupes = opt.noDupes;
    P.noAncestors = opt.noAncestors;
    P.onlyOrthologs = opt.onlyOrthologs;
    P.error = dErr;
    if (opt.targets.empty()) {
        for (int w = 0; w < 4; ++w)
            P.scopeMask[w] = P.targetMask[w] = ~0ull;
    } else {
        // halColumnIterator.cpp:45-51: targets + reference, scope = their spanning tree (halCommon.cpp:156-187)
        std::set<int> tg(opt.targets.begin(), opt.targets.end());
        tg.insert(ref);
        for (int g : tg) {
            if (g < 0 || g >= ng)
                throw std::runtime_error("target genome id out of range");
            P.targetMask[g >> 6] |= 1ull << (g & 63);
        }
        int lca = *tg.begin();
        for (int g : tg)
            lca = img.lca(lca, g);
        std::set<int> scope;
        for (int g : tg)
            for (int x = g;; x = img.genomes[(size_t)x].parent) {
                scope.insert(x);
                if (x == lca)
                    break;
            }
        for (int g : scope)
            P.scopeMask[g >> 6] |= 1ull << (g & 63);
    }
    return P;
}

static constexpr int COL_GRID = 2048;

void columnsDepthDevice(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                        int32_t *d_out, void *stream, ColumnStats *stats) {
    HIP_OK(hipSetDevice(h->dev->device));
    hipStream_t s = (hipStream_t)stream;
    Buf err(4);
    HIP_OK(hipMemsetAsync(err.p, 0, 4, s));
    ColumnParams P = makeParams(h, ref, first, count, step, opt, (unsigned int *)err.p);
    Ev a, b;
    HIP_OK(hipEventRecord(a.e, s));
    if (count > 0) {
        const int grid = (int)std::min<int64_t>(COL_GRID, (count + 255) / 256);
        if (h->dev->wide)
            hipLaunchKernelGGL((k_column_depth<int64_t>), dim3(grid), dim3(256), 0, s, P, mode, d_out);
        else
            hipLaunchKernelGGL((k_column_depth<int32_t>), dim3(grid), dim3(256), 0, s, P, mode, d_out);
    }
    HIP_OK(hipEventRecord(b.e, s));
    unsigned int e = 0;
    HIP_OK(hipMemcpyAsync(&e, err.p, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    if (e)
        throw std::runtime_error("column walk exceeded the frame stack (more than 64 pending branches for one base)");
    if (stats) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a.e, b.e));
        stats->depth_ms += ms;
        stats->columns += (uint64_t)count;
    }
}

void columnsDepthHost(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                      int32_t *out, ColumnStats *stats) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    Buf d((size_t)count * 4);
    columnsDepthDevice(h, ref, first, count, step, mode, opt, (int32_t *)d.p, nullptr, stats);
    if (count > 0)
        HIP_OK(hipMemcpy(out, d.p, (size_t)count * 4, hipMemcpyDeviceToHost));
}

void columnsRowsHost(hgx_alignment *h, int ref, int64_t first, int64_t count, const ColumnOptions &opt, bool withDna,
                     std::vector<uint64_t> &rowOffset, std::vector<ColumnRowHost> &rows, ColumnStats *stats) {
    static_assert(sizeof(ColumnRowHost) == sizeof(ColumnRow), "row layouts must match");
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    if (withDna)
        ensureDeviceDna(h->img, *h->dev);
    // pass 1: rows per column (the same walk, counting); pass 2: emit at the scanned offsets
    std::vector<int32_t> cnt((size_t)count);
    columnsDepthHost(h, ref, first, count, 1, 2, opt, cnt.data(), stats);
    rowOffset.assign((size_t)count + 1, 0);
    for (int64_t i = 0; i < count; ++i)
        rowOffset[(size_t)i + 1] = rowOffset[(size_t)i] + (uint64_t)cnt[(size_t)i];
    const uint64_t total = rowOffset[(size_t)count];
    rows.resize(total);
    if (count == 0)
        return;
    Buf dOff(((size_t)count + 1) * 8), dRows(total * sizeof(ColumnRow)), err(4);
    HIP_OK(hipMemcpy(dOff.p, rowOffset.data(), ((size_t)count + 1) * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(err.p, 0, 4));
    ColumnParams P = makeParams(h, ref, first, count, 1, opt, (unsigned int *)err.p);
    Ev a, b;
    HIP_OK(hipEventRecord(a.e, nullptr));
    const int grid = (int)std::min<int64_t>(COL_GRID, (count + 255) / 256);
    if (h->dev->wide)
        hipLaunchKernelGGL((k_column_rows<int64_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint64_t *)dOff.p, (ColumnRow *)dRows.p);
    else
        hipLaunchKernelGGL((k_column_rows<int32_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint64_t *)dOff.p, (ColumnRow *)dRows.p);
    HIP_OK(hipEventRecord(b.e, nullptr));
    unsigned int e = 0;
    HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
    if (e)
        throw std::runtime_error("column walk exceeded the frame stack (more than 64 pending branches for one base)");
    if (total)
        HIP_OK(hipMemcpy(rows.data(), dRows.p, total * sizeof(ColumnRow), hipMemcpyDeviceToHost));
    if (stats) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a.e, b.e));
        stats->rows_ms += ms;
        stats->rows += total;
    }
}

} // namespace hgx
