#include "hgx_textmem.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <sys/mman.h>

namespace hgx {

namespace {
constexpr size_t kMapFrom = (size_t)1 << 20;   // smaller blocks come from malloc
constexpr size_t kKeepBytes = (size_t)1 << 30; // released blocks kept for reuse: at most this much, in at most two blocks
struct TextMem {
    std::mutex mu;
    std::map<const void *, size_t> live; // mapped blocks handed out: their mapped length
    std::map<const void *, size_t> kept; // released, still mapped
    std::vector<const void *> keptOrder; // ... in the order they were released
    size_t keptBytes = 0;
};
TextMem &mem() {
    static TextMem *m = new TextMem; // (never destroyed: texts may be released from static destructors)
    return *m;
}
size_t pages(size_t bytes) {
    return (bytes + 4095) & ~(size_t)4095;
}
bool adviseHugePages() {
    static const bool advise = !(getenv("HGX_TEXT_HUGEPAGES") && atoi(getenv("HGX_TEXT_HUGEPAGES")) == 0);
    return advise;
}
void *mapFresh(size_t len) {
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED)
        return nullptr;
    // (advice: the 2 MB extents inside the block fault in as one page each.  Where memory is fragmented and transparent_hugepage/defrag
    // is "madvise", such a fault waits for the kernel to compact memory — seconds for a few hundred megabytes on a long-running VM:
    // HGX_TEXT_HUGEPAGES=0 leaves the advice out)
    if (adviseHugePages())
        (void)madvise(p, len, MADV_HUGEPAGE);
    return p;
}
} // namespace

void *textAlloc(size_t bytes) {
    if (bytes < kMapFrom) {
        // (small texts: malloc's; told apart from the mapped ones by the registry)
        return malloc(bytes ? bytes : 1);
    }
    const size_t len = pages(bytes);
    TextMem &M = mem();
    {
        std::lock_guard<std::mutex> lock(M.mu);
        // a kept block that holds the text without being several times its size
        const void *best = nullptr;
        size_t bestLen = 0;
        for (auto &kv : M.kept)
            if (kv.second >= len && kv.second <= 4 * len + ((size_t)16 << 20) && (!best || kv.second < bestLen)) {
                best = kv.first;
                bestLen = kv.second;
            }
        if (best) {
            M.kept.erase(best);
            M.keptOrder.erase(std::find(M.keptOrder.begin(), M.keptOrder.end(), best));
            M.keptBytes -= bestLen;
            M.live[best] = bestLen;
            return const_cast<void *>(best);
        }
    }
    void *p = mapFresh(len);
    if (!p)
        return nullptr;
    std::lock_guard<std::mutex> lock(M.mu);
    M.live[p] = len;
    return p;
}

void *textRealloc(void *p, size_t bytes) {
    if (!p)
        return textAlloc(bytes);
    TextMem &M = mem();
    const size_t len = pages(bytes);
    size_t have = 0;
    {
        // The block leaves the registry BEFORE it is moved (round 6).  mremap gives the block's old addresses back to the kernel, and
        // another thread's mapping may be handed exactly those before this thread is back under the lock: erasing p afterwards
        // erased the OTHER thread's block, whose next textRealloc / textFree then took it for malloc's — "realloc(): invalid
        // pointer", or a GPU copy into pages that had been unmapped: hgx_maf_export_multi's slices, several at a time a handle,
        // lost about every third run of config 3's leg to it (profiles/r06_notes.md 12).
        std::lock_guard<std::mutex> lock(M.mu);
        auto it = M.live.find(p);
        if (it != M.live.end()) {
            have = it->second;
            if (len <= have)
                return p;
            M.live.erase(it);
        }
    }
    if (!have) // malloc's (a caller whose text outgrows malloc moves it over itself: the size of a malloc block is not known here)
        return realloc(p, bytes ? bytes : 1);
    void *q = mremap(p, have, len, MREMAP_MAYMOVE);
    if (q == MAP_FAILED) {
        std::lock_guard<std::mutex> lock(M.mu);
        M.live[p] = have; // (still the caller's, where it was)
        return nullptr;
    }
    if (adviseHugePages())
        (void)madvise(q, len, MADV_HUGEPAGE);
    std::lock_guard<std::mutex> lock(M.mu);
    M.live[q] = len;
    return q;
}

bool textOwns(const void *p) {
    if (!p)
        return false;
    TextMem &M = mem();
    std::lock_guard<std::mutex> lock(M.mu);
    return M.live.count(p) != 0;
}

void textFree(void *p) {
    if (!p)
        return;
    TextMem &M = mem();
    size_t len = 0;
    std::vector<std::pair<void *, size_t>> drop; // (unmapped outside the lock)
    {
        std::lock_guard<std::mutex> lock(M.mu);
        auto it = M.live.find(p);
        if (it != M.live.end()) {
            len = it->second;
            M.live.erase(it);
            // the block just released is the likeliest size of the next text: it is kept, and older ones make room for it (kept
            // in the order they were released: a full set of two small blocks used to turn every larger text away — a fresh
            // mapping of 140 MB and its page faults per call)
            if (len <= kKeepBytes) {
                M.keptOrder.push_back(p);
                M.kept[p] = len;
                M.keptBytes += len;
                while (M.keptOrder.size() > 2 || M.keptBytes > kKeepBytes) {
                    const void *old = M.keptOrder.front();
                    M.keptOrder.erase(M.keptOrder.begin());
                    auto k = M.kept.find(old);
                    M.keptBytes -= k->second;
                    drop.emplace_back(const_cast<void *>(old), k->second);
                    M.kept.erase(k);
                }
            } else {
                drop.emplace_back(p, len);
            }
        }
    }
    if (!len)
        free(p); // malloc's
    for (auto &d : drop)
        (void)munmap(d.first, d.second);
}

void textTrim() {
    TextMem &M = mem();
    std::map<const void *, size_t> drop;
    {
        std::lock_guard<std::mutex> lock(M.mu);
        drop.swap(M.kept);
        M.keptOrder.clear();
        M.keptBytes = 0;
    }
    for (auto &kv : drop)
        (void)munmap(const_cast<void *>(kv.first), kv.second);
}

} // namespace hgx
