// Column engine: launches of the per-base closure kernels behind hgx_columns_depth / hgx_alignment_depth /
// hgx_maf_export (include/hgx.h).
#include "hgx_column_kernels.hpp"
#include "hgx_liftover_kernels.hpp"
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
        hipLaunchKernelGGL((k_column_rows<int64_t, uint64_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint64_t *)dOff.p,
                           (ColumnRow *)dRows.p);
    else
        hipLaunchKernelGGL((k_column_rows<int32_t, uint64_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint64_t *)dOff.p,
                           (ColumnRow *)dRows.p);
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

// exclusive scan of n uint32 on the device (out[n] = total); scratch: (n / 1024 + 2) uint32
static uint32_t deviceScan(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *blockSums) {
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, nullptr, in, n, blockSums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, nullptr, blockSums, nb, out + n);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, nullptr, in, n, (const uint32_t *)blockSums, out);
    uint32_t total = 0;
    HIP_OK(hipMemcpy(&total, out + n, 4, hipMemcpyDeviceToHost));
    return total;
}

void columnsHeadRowsHost(hgx_alignment *h, int ref, int64_t first, int64_t count, const ColumnOptions &opt, bool withDna,
                         std::vector<uint8_t> &head, std::vector<uint32_t> &headOffset, std::vector<ColumnRowHost> &headRows,
                         ColumnStats *stats) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    if (withDna)
        ensureDeviceDna(h->img, *h->dev);
    head.clear();
    headOffset.assign(1, 0);
    headRows.clear();
    if (count == 0)
        return;
    if (count >= ((int64_t)1 << 31))
        throw std::runtime_error("column chunk too large");
    const uint32_t n = (uint32_t)count;
    Buf dCnt((size_t)n * 4), dOff(((size_t)n + 1) * 4), dSums(((size_t)n / SCAN_BLOCK + 2) * 4), err(4);
    Ev e0, e1;
    HIP_OK(hipEventRecord(e0.e, nullptr));
    // 1. rows per column, 2. offsets, 3. all rows (device only)
    columnsDepthDevice(h, ref, first, count, 1, 2, opt, (int32_t *)dCnt.p, nullptr, nullptr);
    const uint32_t totalRows = deviceScan((const uint32_t *)dCnt.p, n, (uint32_t *)dOff.p, (uint32_t *)dSums.p);
    Buf dRows((size_t)totalRows * sizeof(ColumnRow));
    HIP_OK(hipMemset(err.p, 0, 4));
    ColumnParams P = makeParams(h, ref, first, count, 1, opt, (unsigned int *)err.p);
    const int grid = (int)std::min<int64_t>(COL_GRID, (count + 255) / 256);
    if (h->dev->wide)
        hipLaunchKernelGGL((k_column_rows<int64_t, uint32_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint32_t *)dOff.p,
                           (ColumnRow *)dRows.p);
    else
        hipLaunchKernelGGL((k_column_rows<int32_t, uint32_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint32_t *)dOff.p,
                           (ColumnRow *)dRows.p);
    // 4. run heads, 5. offsets of the heads' rows, 6. gather them
    Buf dHead(n), dHeadCnt((size_t)n * 4), dHeadOff(((size_t)n + 1) * 4);
    hipLaunchKernelGGL(k_column_heads, dim3(grid), dim3(256), 0, nullptr, (const uint32_t *)dOff.p, (const ColumnRow *)dRows.p, count,
                       (uint8_t *)dHead.p, (uint32_t *)dHeadCnt.p);
    const uint32_t totalHeadRows = deviceScan((const uint32_t *)dHeadCnt.p, n, (uint32_t *)dHeadOff.p, (uint32_t *)dSums.p);
    Buf dOut((size_t)totalHeadRows * sizeof(ColumnRow));
    hipLaunchKernelGGL(k_gather_head_rows, dim3(grid), dim3(256), 0, nullptr, (const uint32_t *)dOff.p, (const ColumnRow *)dRows.p, count,
                       (const uint8_t *)dHead.p, (const uint32_t *)dHeadOff.p, (ColumnRow *)dOut.p);
    HIP_OK(hipEventRecord(e1.e, nullptr));
    unsigned int e = 0;
    HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
    if (e)
        throw std::runtime_error("column walk exceeded the frame stack (more than 64 pending branches for one base)");
    head.resize(n);
    HIP_OK(hipMemcpy(head.data(), dHead.p, n, hipMemcpyDeviceToHost));
    std::vector<uint32_t> headCnt(n);
    HIP_OK(hipMemcpy(headCnt.data(), dHeadCnt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    headRows.resize(totalHeadRows);
    if (totalHeadRows)
        HIP_OK(hipMemcpy(headRows.data(), dOut.p, (size_t)totalHeadRows * sizeof(ColumnRow), hipMemcpyDeviceToHost));
    uint32_t acc = 0;
    for (uint32_t c = 0; c < n; ++c)
        if (head[c]) {
            acc += headCnt[c];
            headOffset.push_back(acc);
        }
    if (stats) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0.e, e1.e));
        stats->rows_ms += ms;
        stats->rows += totalRows;
        stats->columns += (uint64_t)count;
    }
}

} // namespace hgx
