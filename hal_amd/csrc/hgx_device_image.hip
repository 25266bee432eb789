// The device image and the workspace cache behind every handle (hgx_device.hpp): uploading an alignment's tables to HBM, the
// per-device cache of released workspace blocks, the image's release.  Plain HIP API calls, no kernels: this translation unit is
// also what the host-side emulation of the column engine links (tests/cpp/hipshim: the same upload code over host memory).
#include "hgx_device.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace hgx {

#define HIP_OK(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// workspace cache (hgx_device.hpp)
namespace {
struct DevCache {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void *>> free; // (device, size class) -> released blocks
    std::map<void *, std::pair<int, size_t>> live;              // every block handed out: its device and size class
    std::map<int, size_t> cachedBytes;
};
DevCache &devCache() {
    static DevCache *c = new DevCache; // (never destroyed: blocks may be released from static destructors)
    return *c;
}
size_t sizeClass(size_t bytes) {
    // (above 256 MiB the classes are 64 MiB apart: a 9 GB workspace must not become a 12 GB one)
    if (bytes > ((size_t)256 << 20))
        return (bytes + (((size_t)64 << 20) - 1)) & ~(((size_t)64 << 20) - 1);
    size_t c = 256;
    while (c < bytes) {
        if (c + c / 2 >= bytes && c >= 4096)
            return c + c / 2;
        c <<= 1;
    }
    return c;
}
} // namespace

void *devAlloc(size_t bytes) {
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    const size_t cls = sizeClass(std::max<size_t>(bytes, 16));
    DevCache &C = devCache();
    {
        std::lock_guard<std::mutex> lock(C.mu);
        auto it = C.free.find({dev, cls});
        if (it != C.free.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            C.cachedBytes[dev] -= cls;
            C.live[p] = {dev, cls};
            return p;
        }
    }
    void *p = nullptr;
    size_t got = cls;
    hipError_t e = hipMalloc(&p, cls);
    if (e != hipSuccess) { // out of memory with blocks in the cache: give them back and try once more
        (void)hipGetLastError();
        devCacheTrim(dev);
        e = hipMalloc(&p, cls);
    }
    if (e != hipSuccess && bytes < cls) { // the class does not fit, the request itself may: an exact block (its own class when it comes back)
        (void)hipGetLastError();
        got = std::max<size_t>(bytes, 16);
        e = hipMalloc(&p, got);
    }
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " at hipMalloc of " + std::to_string(bytes) + " bytes");
    std::lock_guard<std::mutex> lock(C.mu);
    C.live[p] = {dev, got};
    return p;
}

void devRelease(void *p) {
    if (!p)
        return;
    static const size_t limit = []() {
        const char *e = getenv("HGX_CACHE_BYTES");
        return e ? (size_t)std::max<long long>(0, atoll(e)) : (size_t)24 << 30;
    }();
    DevCache &C = devCache();
    int dev = -1;
    size_t cls = 0;
    bool keep = false;
    {
        std::lock_guard<std::mutex> lock(C.mu);
        auto it = C.live.find(p);
        if (it != C.live.end()) {
            dev = it->second.first;
            cls = it->second.second;
            C.live.erase(it);
            keep = C.cachedBytes[dev] + cls <= limit;
            if (keep) {
                C.free[{dev, cls}].push_back(p);
                C.cachedBytes[dev] += cls;
            }
        }
    }
    if (!keep) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (dev >= 0 && dev != cur)
            (void)hipSetDevice(dev);
        (void)hipFree(p);
        if (dev >= 0 && dev != cur)
            (void)hipSetDevice(cur);
    }
}

void devCacheTrim(int device) {
    DevCache &C = devCache();
    std::vector<void *> blocks;
    {
        std::lock_guard<std::mutex> lock(C.mu);
        for (auto &kv : C.free)
            if (kv.first.first == device) {
                blocks.insert(blocks.end(), kv.second.begin(), kv.second.end());
                kv.second.clear();
            }
        C.cachedBytes[device] = 0;
    }
    if (blocks.empty())
        return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != device)
        (void)hipSetDevice(device);
    for (void *p : blocks)
        (void)hipFree(p);
    if (cur != device)
        (void)hipSetDevice(cur);
}

// ---------------------------------------------------------------------------------------------
// device image
DeviceImage::~DeviceImage() {
    if (device < 0)
        return;
    (void)hipSetDevice(device);
    devCacheTrim(device);
    for (DeviceGenome &g : genomes) {
        if (g.top)
            (void)hipFree(g.top);
        if (g.up)
            (void)hipFree(g.up);
        if (g.chainMid)
            (void)hipFree(g.chainMid);
        if (g.chainLast)
            (void)hipFree(g.chainLast);
        if (g.bot)
            (void)hipFree(g.bot);
        for (int32_t *c : g.childEnc)
            if (c)
                (void)hipFree(c);
        for (void *c : g.downRec)
            if (c)
                (void)hipFree(c);
        for (int32_t *c : g.locate)
            if (c)
                (void)hipFree(c);
        if (g.seqStart)
            (void)hipFree(g.seqStart);
    }
    for (auto &kv : composed) {
        if (kv.second.recs)
            (void)hipFree(kv.second.recs);
        if (kv.second.eo)
            (void)hipFree(kv.second.eo);
        if (kv.second.coarse)
            (void)hipFree(kv.second.coarse);
        if (kv.second.starts)
            (void)hipFree(kv.second.starts);
        if (kv.second.mRecs)
            (void)hipFree(kv.second.mRecs);
        if (kv.second.mBuckets)
            (void)hipFree(kv.second.mBuckets);
        if (kv.second.mFlagBits)
            (void)hipFree(kv.second.mFlagBits);
    }
    if (desc)
        (void)hipFree(desc);
    if (childPtrs)
        (void)hipFree((void *)childPtrs);
    if (childGenomes)
        (void)hipFree(childGenomes);
    for (uint8_t *p : dna)
        if (p)
            (void)hipFree(p);
}

static void uploadDescs(const Image &img, DeviceImage &D) {
    std::vector<GenomeDesc> descs(img.genomes.size());
    std::vector<const int32_t *> ptrs;
    std::vector<int32_t> ids;
    std::vector<size_t> firstChild(img.genomes.size());
    for (size_t g = 0; g < img.genomes.size(); ++g) {
        firstChild[g] = ptrs.size();
        for (size_t k = 0; k < img.genomes[g].children.size(); ++k) {
            ptrs.push_back(D.genomes[g].childEnc[k]);
            ids.push_back(img.genomes[g].children[k]);
        }
    }
    if (!D.childPtrs) {
        HIP_OK(hipMalloc((void **)&D.childPtrs, std::max<size_t>(1, ptrs.size()) * sizeof(int32_t *)));
        HIP_OK(hipMalloc((void **)&D.childGenomes, std::max<size_t>(1, ids.size()) * 4));
    }
    if (!ptrs.empty()) {
        HIP_OK(hipMemcpy(D.childPtrs, ptrs.data(), ptrs.size() * sizeof(int32_t *), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(D.childGenomes, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    }
    for (size_t g = 0; g < img.genomes.size(); ++g) {
        const GenomeTables &G = img.genomes[g];
        GenomeDesc &d = descs[g];
        memset(&d, 0, sizeof d);
        d.top = D.genomes[g].top;
        d.bot = D.genomes[g].bot;
        d.child = (const int32_t *const *)(D.childPtrs + firstChild[g]);
        d.childGenome = D.childGenomes + firstChild[g];
        d.dna = g < D.dna.size() ? D.dna[g] : nullptr;
        d.seqStart = D.genomes[g].seqStart;
        d.numTop = G.numTop;
        d.numBot = G.numBot;
        d.length = G.totalLength;
        d.parent = G.parent;
        d.slotInParent = G.parent >= 0 ? img.genomes[(size_t)G.parent].childSlotOf((int)g) : -1;
        d.numChildren = (int32_t)G.children.size();
        d.numSeq = (int32_t)G.seqs.size();
    }
    if (!D.desc)
        HIP_OK(hipMalloc(&D.desc, std::max<size_t>(1, descs.size()) * sizeof(GenomeDesc)));
    HIP_OK(hipMemcpy(D.desc, descs.data(), descs.size() * sizeof(GenomeDesc), hipMemcpyHostToDevice));
}

void ensureDeviceDna(const Image &img, DeviceImage &D) {
    static std::mutex mu; // (several exports at a time on one handle: hgx_maf_export_multi's slices)
    std::lock_guard<std::mutex> lock(mu);
    if (!D.dna.empty())
        return;
    HIP_OK(hipSetDevice(D.device));
    D.dna.assign(img.genomes.size(), nullptr);
    for (size_t g = 0; g < img.genomes.size(); ++g) {
        const std::vector<uint8_t> &p = img.genomes[g].dna;
        if (p.empty())
            continue;
        HIP_OK(hipMalloc(&D.dna[g], p.size()));
        HIP_OK(hipMemcpy(D.dna[g], p.data(), p.size(), hipMemcpyHostToDevice));
        D.bytes += p.size();
    }
    uploadDescs(img, D);
}

static inline int32_t encLink(int64_t idx, bool rev) {
    return idx < 0 ? -1 : (int32_t)((idx << 1) | (rev ? 1 : 0));
}

template <typename C> static void uploadUpTable(const GenomeTables &G, const GenomeTables &P, DeviceGenome &D, size_t &bytes) {
    std::vector<UpRec<C>> up((size_t)G.numTop + 1);
    memset(up.data(), 0, up.size() * sizeof(UpRec<C>));
    for (int64_t i = 0; i < G.numTop; ++i) {
        UpRec<C> &r = up[(size_t)i];
        const int64_t p = G.tParent[(size_t)i];
        r.start = (C)G.tStart[(size_t)i];
        r.parentEnc = p < 0 ? -1 : (int32_t)((p << 1) | (G.tParentRev[(size_t)i] ? 1 : 0));
        r.parentStart = p < 0 ? 0 : (C)P.bStart[(size_t)p];
        r.parentTopParse = p < 0 ? -1 : (int32_t)P.bTopParse[(size_t)p];
    }
    up[(size_t)G.numTop].start = (C)G.totalLength;
    up[(size_t)G.numTop].parentEnc = -1;
    up[(size_t)G.numTop].parentTopParse = -1;
    HIP_OK(hipMalloc(&D.up, up.size() * sizeof(UpRec<C>)));
    HIP_OK(hipMemcpy(D.up, up.data(), up.size() * sizeof(UpRec<C>), hipMemcpyHostToDevice));
    bytes += up.size() * sizeof(UpRec<C>);
}

template <typename C> static void uploadGenome(const GenomeTables &G, DeviceGenome &D, size_t &bytes) {
    std::vector<TopRec<C>> top((size_t)G.numTop + 1);
    memset(top.data(), 0, top.size() * sizeof(TopRec<C>));
    for (int64_t i = 0; i < G.numTop; ++i) {
        TopRec<C> &r = top[(size_t)i];
        r.start = (C)G.tStart[(size_t)i];
        r.parentEnc = encLink(G.tParent[(size_t)i], G.tParentRev[(size_t)i] != 0);
        r.paralogy = (int32_t)G.tParalogy[(size_t)i];
        r.botParse = (int32_t)G.tBotParse[(size_t)i];
    }
    top[(size_t)G.numTop].start = (C)G.totalLength;
    top[(size_t)G.numTop].parentEnc = -1;
    top[(size_t)G.numTop].paralogy = -1;
    top[(size_t)G.numTop].botParse = -1;
    std::vector<BotRec<C>> bot((size_t)G.numBot + 1);
    memset(bot.data(), 0, bot.size() * sizeof(BotRec<C>));
    for (int64_t i = 0; i < G.numBot; ++i) {
        bot[(size_t)i].start = (C)G.bStart[(size_t)i];
        bot[(size_t)i].topParse = (int32_t)G.bTopParse[(size_t)i];
    }
    bot[(size_t)G.numBot].start = (C)G.totalLength;
    bot[(size_t)G.numBot].topParse = -1;
    HIP_OK(hipMalloc(&D.top, top.size() * sizeof(TopRec<C>)));
    HIP_OK(hipMemcpy(D.top, top.data(), top.size() * sizeof(TopRec<C>), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&D.bot, bot.size() * sizeof(BotRec<C>)));
    HIP_OK(hipMemcpy(D.bot, bot.data(), bot.size() * sizeof(BotRec<C>), hipMemcpyHostToDevice));
    bytes += top.size() * sizeof(TopRec<C>) + bot.size() * sizeof(BotRec<C>);
}

std::unique_ptr<DeviceImage> uploadImage(const Image &img, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw std::runtime_error("no HIP device available: the liftover/column kernels require a GPU (gfx950)");
    if (device < 0 || device >= ndev)
        throw std::runtime_error("invalid HIP device ordinal " + std::to_string(device));
    HIP_OK(hipSetDevice(device));
    std::unique_ptr<DeviceImage> D(new DeviceImage);
    D->device = device;
    for (const GenomeTables &G : img.genomes) {
        if (G.totalLength >= (int64_t)1 << 31)
            D->wide = true;
        if (G.numTop >= ((int64_t)1 << 30) - 1 || G.numBot >= ((int64_t)1 << 30) - 1)
            throw std::runtime_error("genome " + G.name + " has more than 2^30 segments; 32-bit link tables cannot hold it");
        for (int64_t i = 0; i < G.numTop; ++i)
            if (G.tStart[(size_t)i + 1] - G.tStart[(size_t)i] >= (int64_t)1 << 31)
                throw std::runtime_error("genome " + G.name + " has a segment of 2^31 bases or more");
    }
    // HGX_FORCE_WIDE=1 selects the int64 coordinate tables regardless of genome size (exercises the path that
    // genomes of 2^31 bases or more take)
    if (const char *fw = getenv("HGX_FORCE_WIDE"))
        if (fw[0] == '1')
            D->wide = true;
    D->genomes.resize(img.genomes.size());
    for (size_t g = 0; g < img.genomes.size(); ++g) {
        const GenomeTables &G = img.genomes[g];
        DeviceGenome &dg = D->genomes[g];
        dg.numTop = G.numTop;
        dg.numBot = G.numBot;
        if (D->wide)
            uploadGenome<int64_t>(G, dg, D->bytes);
        else
            uploadGenome<int32_t>(G, dg, D->bytes);
        if (G.parent >= 0 && G.numTop > 0) {
            if (D->wide)
                uploadUpTable<int64_t>(G, img.genomes[(size_t)G.parent], dg, D->bytes);
            else
                uploadUpTable<int32_t>(G, img.genomes[(size_t)G.parent], dg, D->bytes);
        }
        dg.childEnc.assign(G.children.size(), nullptr);
        std::vector<int32_t> enc((size_t)G.numBot);
        for (size_t k = 0; k < G.children.size(); ++k) {
            for (int64_t i = 0; i < G.numBot; ++i)
                enc[(size_t)i] = encLink(G.bChild[k][(size_t)i], G.bChildRev[k][(size_t)i] != 0);
            HIP_OK(hipMalloc(&dg.childEnc[k], std::max<size_t>(4, enc.size() * 4)));
            HIP_OK(hipMemcpy(dg.childEnc[k], enc.data(), enc.size() * 4, hipMemcpyHostToDevice));
            D->bytes += enc.size() * 4;
        }
        std::vector<int64_t> ss;
        for (const SeqInfo &S : G.seqs)
            ss.push_back(S.start);
        ss.push_back(G.totalLength);
        dg.numSeq = (int32_t)G.seqs.size();
        HIP_OK(hipMalloc(&dg.seqStart, ss.size() * 8));
        HIP_OK(hipMemcpy(dg.seqStart, ss.data(), ss.size() * 8, hipMemcpyHostToDevice));
    }
    uploadDescs(img, *D);
    return D;
}

} // namespace hgx
