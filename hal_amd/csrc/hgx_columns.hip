// Column engine: launches of the per-base closure kernels behind hgx_columns_depth / hgx_alignment_depth /
// hgx_maf_export (include/hgx.h).
#include "hgx_column_kernels.hpp"
#include "hgx_gap_kernels.hpp"
#include "hgx_maf_kernels.hpp"
#include "hgx_maf_render_kernels.hpp"
#include "hgx_scan_kernels.hpp"
#include "hgx_columns_engine.hpp"
#include "hgx_liftover_engine.hpp"
#include "hgx_wig_text.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
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
    Buf() = default;
    explicit Buf(size_t bytes) {
        p = hgx::devAlloc(bytes);
    }
    void resize(size_t bytes) { // contents are not kept
        hgx::devRelease(p);
        p = nullptr;
        p = hgx::devAlloc(bytes);
    }
    ~Buf() {
        hgx::devRelease(p);
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

// masks: device memory of the call that holds the scope and target bit sets (ceil(genomes / 64) words each)
static ColumnParams makeParams(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, const ColumnOptions &opt,
                               unsigned int *dErr, Buf &masks) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    const Image &img = h->img;
    const int ng = (int)img.genomes.size();
    if (ref < 0 || ref >= ng)
        throw std::runtime_error("reference genome id out of range");
    if (ng > 2048)
        throw std::runtime_error("alignments with more than 2048 genomes are not supported by the column kernels (the depth kernels keep a 2048-bit genome set per column)");
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
    P.derefs = nullptr;
    const size_t words = ((size_t)ng + 63) / 64;
    std::vector<unsigned long long> m(2 * words, 0ull); // scope, then targets
    if (opt.targets.empty()) {
        std::fill(m.begin(), m.end(), ~0ull);
    } else {
        // halColumnIterator.cpp:45-51: targets + reference, scope = their spanning tree (halCommon.cpp:156-187)
        std::set<int> tg(opt.targets.begin(), opt.targets.end());
        tg.insert(ref);
        for (int g : tg) {
            if (g < 0 || g >= ng)
                throw std::runtime_error("target genome id out of range");
            m[words + (size_t)(g >> 6)] |= 1ull << (g & 63);
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
            m[(size_t)(g >> 6)] |= 1ull << (g & 63);
    }
    masks.resize(16 * words);
    HIP_OK(hipMemcpy(masks.p, m.data(), 16 * words, hipMemcpyHostToDevice));
    P.scopeMask = (const unsigned long long *)masks.p;
    P.targetMask = (const unsigned long long *)masks.p + words;
    return P;
}

static const int COL_GRID = getenv("HGX_COL_GRID") ? atoi(getenv("HGX_COL_GRID")) : 2048;

static bool perBaseColumns() { // (looked at on every call: tests compare the two depth kernels inside one process)
    return getenv("HGX_COLUMNS_PER_BASE") != nullptr;
}

// reference segments (top tiling, or bottom for a genome without one) holding positions firstPos .. lastPos
static void refSegmentRange(hgx_alignment *h, int ref, int64_t firstPos, int64_t lastPos, int32_t &segFirst, int64_t &segCount) {
    const GenomeTables &G = h->img.genomes[(size_t)ref];
    const std::vector<int64_t> &st = G.numTop > 0 ? G.tStart : G.bStart;
    const int64_t nseg = G.numTop > 0 ? G.numTop : G.numBot;
    if (nseg <= 0 || firstPos < 0 || lastPos >= G.totalLength || lastPos < firstPos)
        throw std::runtime_error("column range outside the reference genome");
    auto find = [&](int64_t p) { return (int64_t)(std::upper_bound(st.begin(), st.begin() + nseg, p) - st.begin()) - 1; };
    const int64_t a = find(firstPos), b = find(lastPos);
    segFirst = (int32_t)a;
    segCount = b - a + 1;
}

// the dereference counters cost ~10 % of the kernel, so they are a separate instantiation
// (alignments of more than 256 genomes take the instantiation with a 2048-bit genome set per column)
#define LAUNCH_DEPTH(K, CT, ...)                                                                                       \
    do {                                                                                                               \
        if (P.numGenomes > 256)                                                                                        \
            hipLaunchKernelGGL((K<CT, false, 32>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);                          \
        else if (countDerefs)                                                                                          \
            hipLaunchKernelGGL((K<CT, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);                               \
        else                                                                                                           \
            hipLaunchKernelGGL((K<CT, false>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);                              \
    } while (0)

// halAlignmentDepth's per-column value by two streaming sweeps over the tree (hgx_column_kernels.hpp: k_sweep_up / k_sweep_down)
// instead of one depth-first walk per run of columns: every in-scope genome gets a track with one entry per base.  Used for
// requests that span a million columns or more (the sweeps always cover whole genomes) when at most 64 genomes count and
// the tracks fit the device; otherwise (return false) the column walk runs.  HGX_DEPTH_SWEEP=0 forbids it, =1 lifts the
// size threshold.  mode 0: genomes - 1, 1: bases - 1 (--countDupes), 2: bases.
// what the sweeps need to know of a genome's track: the bytes of a word (1 << wlog), its first genome in the numbering of the
// counted genomes (lo: a track's bit b is genome lo + b), its own bit (or 1 for sums; 0: not counted)
struct SweepTrack {
    int wlog = 0, lo = 0;
    long long own = 0;
};
// workgroups of a sweep launch (4096: 2-3 % slower, profiles/r06x_*; HGX_SWEEP_GRID: experiments, and the host-side emulation of the
// tests, whose launches are loops over the grid's threads)
static int sweepGrid() {
    const char *e = getenv("HGX_SWEEP_GRID");
    return e ? std::max(1, atoi(e)) : 8192;
}
template <typename C, bool SUM, typename AT>
static void sweepTracks(hgx_alignment *h, const std::vector<int> &postOrder, const std::vector<int> &path, const std::vector<char> &inScope,
                        const std::vector<SweepTrack> &track, const std::vector<char> &hasTrack, std::vector<Buf> &S, std::vector<Buf> &A,
                        hipStream_t s, bool noDupes = false, bool topSizes = false) {
    // topSizes (halAlignmentDepth, genome sets): the genome at the top of the scope stores its sets' SIZES, as bytes, in A — nobody
    // reads its sets but to count them (the caller has made sure its children go in one launch, and given it no S)
    const Image &img = h->img;
    const DeviceImage &D = *h->dev;
    const int GRID = sweepGrid();
    // (k_sweep_up_words for the sets of 2, 4 and 8 bytes: HGX_SWEEP_WORDS=0 keeps the generic kernel, element by element)
    const bool wordsAsked = !(getenv("HGX_SWEEP_WORDS") && getenv("HGX_SWEEP_WORDS")[0] == '0') &&
                            !(getenv("HGX_SWEEP_AHEAD") && getenv("HGX_SWEEP_AHEAD")[0] == '0');
    const bool words = !SUM && wordsAsked, wordsSum = SUM && wordsAsked;
    // (--noDupes: the parent's links to a genome of the path, by which k_sweep_down tells the segment that goes up)
    auto linksTo = [&](int p, int c) -> const int32_t * {
        if (!noDupes)
            return nullptr;
        const GenomeTables &P = img.genomes[(size_t)p];
        for (size_t k = 0; k < P.children.size(); ++k)
            if (P.children[k] == c)
                return D.genomes[(size_t)p].childEnc[k];
        return nullptr;
    };
    for (int g : postOrder) {
        const GenomeTables &G = img.genomes[(size_t)g];
        const DeviceGenome &dg = D.genomes[(size_t)g];
        if (G.totalLength <= 0 || !hasTrack[(size_t)g])
            continue; // (a genome without in-scope children has the same value on every base: no track)
        const SweepTrack &tg = track[(size_t)g];
        std::vector<SweepChild> kids;
        for (size_t k = 0; k < G.children.size(); ++k) {
            const int c = G.children[k];
            if (!inScope[(size_t)c] || img.genomes[(size_t)c].totalLength <= 0 || img.genomes[(size_t)c].numTop <= 0)
                continue;
            const SweepTrack &tc = track[(size_t)c];
            const int shift = SUM ? 0 : tc.lo - tg.lo;
            kids.push_back(SweepChild{dg.childEnc[k], D.genomes[(size_t)c].top, hasTrack[(size_t)c] ? S[(size_t)c].p : nullptr,
                                      SUM ? tc.own : (long long)((unsigned long long)tc.own << shift), tc.wlog, shift});
        }
        for (size_t at = 0; at < kids.size(); at += SWEEP_MAX_CHILDREN) {
            SweepChildren ch;
            ch.noRing = noDupes ? 1 : 0;
            ch.n = (int)std::min<size_t>(SWEEP_MAX_CHILDREN, kids.size() - at);
            for (int k = 0; k < ch.n; ++k)
                ch.c[k] = kids[at + (size_t)k];
#define HGX_UP(M)                                                                                                                            \
    hipLaunchKernelGGL((k_sweep_up<C, M, SUM>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)dg.bot, (int64_t)G.numBot, ch, (M)tg.own,     \
                       at ? 1 : 0, (M *)S[(size_t)g].p)
#define HGX_UP_SIZES(M)                                                                                                                      \
    hipLaunchKernelGGL((k_sweep_up<C, M, false, uint8_t>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)dg.bot, (int64_t)G.numBot, ch,    \
                       (M)tg.own, 0, (uint8_t *)A[(size_t)g].p)
            if (!SUM && topSizes && g == path[0]) {
                if (kids.size() > SWEEP_MAX_CHILDREN || sizeof(AT) != 1)
                    throw std::runtime_error("internal: sizes at the top of the scope are one launch's");
#define HGX_UP_WORDS(W, SIZES, OUT)                                                                                                          \
    hipLaunchKernelGGL((k_sweep_up_words<C, W, SIZES>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)dg.bot, (int64_t)G.numBot, ch,        \
                       (unsigned long long)tg.own, at ? 1 : 0, (uint8_t *)(OUT))
                if (tg.wlog == 0)
                    hipLaunchKernelGGL((k_sweep_up_bytes<C, 2, true>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)dg.bot, (int64_t)G.numBot, ch,
                                       (uint8_t)tg.own, 0, (uint8_t *)A[(size_t)g].p);
                else if (!words && tg.wlog == 1)
                    HGX_UP_SIZES(uint16_t);
                else if (!words && tg.wlog == 2)
                    HGX_UP_SIZES(uint32_t);
                else if (!words)
                    HGX_UP_SIZES(unsigned long long);
                else if (tg.wlog == 1)
                    HGX_UP_WORDS(2, true, A[(size_t)g].p);
                else if (tg.wlog == 2)
                    HGX_UP_WORDS(4, true, A[(size_t)g].p);
                else
                    HGX_UP_WORDS(8, true, A[(size_t)g].p);
            } else if (SUM && wordsSum)
                hipLaunchKernelGGL((k_sweep_up_words<C, 4, false, true>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)dg.bot, (int64_t)G.numBot, ch,
                                   (unsigned long long)(uint32_t)tg.own, at ? 1 : 0, (uint8_t *)S[(size_t)g].p);
            else if (SUM)
                HGX_UP(int32_t);
            else if (tg.wlog == 0 && !(getenv("HGX_SWEEP_AHEAD") && getenv("HGX_SWEEP_AHEAD")[0] == '0'))
                hipLaunchKernelGGL((k_sweep_up_bytes<C, 2>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)dg.bot, (int64_t)G.numBot, ch, (uint8_t)tg.own,
                                   at ? 1 : 0, (uint8_t *)S[(size_t)g].p);
            else if (tg.wlog == 0)
                HGX_UP(uint8_t);
            else if (words && tg.wlog == 1)
                HGX_UP_WORDS(2, false, S[(size_t)g].p);
            else if (words && tg.wlog == 2)
                HGX_UP_WORDS(4, false, S[(size_t)g].p);
            else if (words)
                HGX_UP_WORDS(8, false, S[(size_t)g].p);
            else if (tg.wlog == 1)
                HGX_UP(uint16_t);
            else if (tg.wlog == 2)
                HGX_UP(uint32_t);
            else
                HGX_UP(unsigned long long);
#undef HGX_UP
#undef HGX_UP_SIZES
#undef HGX_UP_WORDS
        }
    }
    // top-down along the path from the top of the scope to the reference (the top's own A is the size of its S: read from S)
    auto sizeOfOwn = [&](int g) { return SUM ? (int32_t)track[(size_t)g].own : (int32_t)__builtin_popcountll((unsigned long long)track[(size_t)g].own); };
    const int top = path[0];
    const bool topDone = !SUM && topSizes && hasTrack[(size_t)top]; // (A of the top is there already)
    if (path.size() == 1 && !topDone) {
        const void *St = hasTrack[(size_t)top] ? S[(size_t)top].p : nullptr;
        const int64_t n = (int64_t)img.genomes[(size_t)top].totalLength;
        const SweepTrack &tt = track[(size_t)top];
#define HGX_TOP(M) hipLaunchKernelGGL((k_sweep_top<M, SUM, AT>), dim3(GRID), dim3(256), 0, s, (const M *)St, n, (M)tt.own, (AT *)A[(size_t)top].p)
        if (SUM)
            HGX_TOP(int32_t);
        else if (tt.wlog == 0)
            HGX_TOP(uint8_t);
        else if (tt.wlog == 1)
            HGX_TOP(uint16_t);
        else if (tt.wlog == 2)
            HGX_TOP(uint32_t);
        else
            HGX_TOP(unsigned long long);
#undef HGX_TOP
    }
    for (size_t i = 1; i < path.size(); ++i) {
        const int c = path[i], p = path[i - 1];
        const void *Sc = hasTrack[(size_t)c] ? S[(size_t)c].p : nullptr;
#define HGX_DOWN(M)                                                                                                                          \
    hipLaunchKernelGGL((k_sweep_down<C, M, SUM, AT>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)D.genomes[(size_t)c].top,               \
                       (int64_t)img.genomes[(size_t)c].numTop, (const BotRec<C> *)D.genomes[(size_t)p].bot,                                   \
                       i == 1 ? (const AT *)nullptr : (const AT *)A[(size_t)p].p, i == 1 ? (const M *)S[(size_t)p].p : (const M *)nullptr, Sc,  \
                       track[(size_t)c].wlog, sizeOfOwn(c), (AT *)A[(size_t)c].p, linksTo(p, c))
        const int pl = track[(size_t)p].wlog;
        if (SUM)
            HGX_DOWN(int32_t);
        else if ((i > 1 || topDone) && sizeof(AT) == 1 && (topDone || !(getenv("HGX_SWEEP_AHEAD") && getenv("HGX_SWEEP_AHEAD")[0] == '0')))
            hipLaunchKernelGGL((k_sweep_down_bytes<C>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)D.genomes[(size_t)c].top,
                               (int64_t)img.genomes[(size_t)c].numTop, (const BotRec<C> *)D.genomes[(size_t)p].bot, (const uint8_t *)A[(size_t)p].p, Sc,
                               track[(size_t)c].wlog, sizeOfOwn(c), (uint8_t *)A[(size_t)c].p, linksTo(p, c));
        else if (i > 1 || pl == 0) // (the parent's track is read by the first step only)
            HGX_DOWN(uint8_t);
        else if (pl == 1)
            HGX_DOWN(uint16_t);
        else if (pl == 2)
            HGX_DOWN(uint32_t);
        else
            HGX_DOWN(unsigned long long);
#undef HGX_DOWN
    }
}

// the walk's scope (halColumnIterator.cpp:45-51), the genomes whose bases are reported (:802-812), a post-order of the scope's tree
// and the path from its top to the reference: what the tree sweeps go by
struct SweepScope {
    std::vector<char> inScope, counted;
    int scopeRoot = -1;
    std::vector<int> postOrder, path;
};
static SweepScope sweepScope(const Image &img, int ref, const ColumnOptions &opt) {
    const int ng = (int)img.genomes.size();
    SweepScope sc;
    sc.inScope.assign((size_t)ng, 1);
    sc.counted.assign((size_t)ng, 1);
    std::vector<char> &inScope = sc.inScope, &counted = sc.counted;
    int scopeRoot = img.root();
    if (!opt.targets.empty()) {
        std::set<int> tg(opt.targets.begin(), opt.targets.end());
        tg.insert(ref);
        scopeRoot = ref;
        for (int g : tg) {
            if (g < 0 || g >= ng)
                throw std::runtime_error("target genome id out of range");
            scopeRoot = img.lca(scopeRoot, g);
        }
        std::fill(inScope.begin(), inScope.end(), 0);
        std::fill(counted.begin(), counted.end(), 0);
        for (int g : tg) {
            counted[(size_t)g] = 1;
            for (int x = g;; x = img.genomes[(size_t)x].parent) {
                inScope[(size_t)x] = 1;
                if (x == scopeRoot)
                    break;
            }
        }
    }
    if (opt.noAncestors)
        for (int g = 0; g < ng; ++g)
            if (!img.genomes[(size_t)g].children.empty())
                counted[(size_t)g] = 0;
    // post-order of the scope's tree, the path to the reference
    std::vector<int> &postOrder = sc.postOrder, &path = sc.path, stack{scopeRoot};
    while (!stack.empty()) { // reverse pre-order = a post-order
        const int g = stack.back();
        stack.pop_back();
        postOrder.push_back(g);
        for (int c : img.genomes[(size_t)g].children)
            if (inScope[(size_t)c])
                stack.push_back(c);
    }
    std::reverse(postOrder.begin(), postOrder.end());
    for (int x = ref;; x = img.genomes[(size_t)x].parent) {
        path.push_back(x);
        if (x == scopeRoot)
            break;
    }
    std::reverse(path.begin(), path.end());
    sc.scopeRoot = scopeRoot;
    return sc;
}

static bool columnsDepthSweep(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                              int32_t *d_out, hipStream_t s, ColumnStats *stats) {
    const char *env = getenv("HGX_DEPTH_SWEEP");
    if (count <= 0 || (env && env[0] == '0') || opt.noDupes || opt.onlyOrthologs)
        return false;
    const int64_t span = (count - 1) * step + 1;
    if (span < (1 << 20) && !(env && env[0] == '1'))
        return false;
    const Image &img = h->img;
    const int ng = (int)img.genomes.size();
    const SweepScope sc = sweepScope(img, ref, opt);
    const std::vector<char> &inScope = sc.inScope, &counted = sc.counted;
    const std::vector<int> &postOrder = sc.postOrder, &path = sc.path;
    const bool sum = mode != 0;
    int bits = 0;
    std::vector<int> bitOf((size_t)ng, -1); // a counted genome's number among the counted ones: post-order, so a subtree's are consecutive
    std::vector<int> firstBit((size_t)ng, 0), below((size_t)ng, 0); // a subtree's first number, and how many it holds
    for (int g : postOrder) {
        int f = bits, n = 0;
        bool any = false;
        for (int c : img.genomes[(size_t)g].children)
            if (inScope[(size_t)c]) {
                if (!any || firstBit[(size_t)c] < f)
                    f = firstBit[(size_t)c];
                any = true;
                n += below[(size_t)c];
            }
        if (counted[(size_t)g]) {
            bitOf[(size_t)g] = bits++;
            ++n;
        }
        firstBit[(size_t)g] = any ? f : (counted[(size_t)g] ? bitOf[(size_t)g] : bits);
        below[(size_t)g] = n;
    }
    // more than 64 counted genomes (cactus alignments have hundreds): the sweeps run once per group of 64 — the size of a
    // column's genome set is the sum of the sizes of its parts — and the groups' depths are added up on the way out
    const int groups = sum ? 1 : std::max(1, (bits + 63) / 64);
    if (getenv("HGX_SWEEP_GROUPS_MAX") && groups > atoi(getenv("HGX_SWEEP_GROUPS_MAX")))
        return false;
    // A genome's set holds the genomes of its subtree only, in a word just wide enough (HGX_SWEEP_LOCAL=0, and alignments that
    // go in groups: one numbering and one width for all)
    const bool local = !sum && groups == 1 && !(getenv("HGX_SWEEP_LOCAL") && getenv("HGX_SWEEP_LOCAL")[0] == '0');
    const int globalLog = sum ? 2 : bits <= 8 ? 0 : bits <= 16 ? 1 : bits <= 32 ? 2 : 3;
    std::vector<SweepTrack> track((size_t)ng);
    auto setTracks = [&](int group) {
        for (int g = 0; g < ng; ++g) {
            SweepTrack &t = track[(size_t)g];
            const int b = bitOf[(size_t)g];
            if (sum) {
                t = SweepTrack{2, 0, b >= 0 ? 1ll : 0ll};
            } else if (local) {
                const int n = below[(size_t)g];
                t.wlog = n <= 8 ? 0 : n <= 16 ? 1 : n <= 32 ? 2 : 3;
                t.lo = firstBit[(size_t)g];
                t.own = b >= 0 ? (long long)(1ull << (b - t.lo)) : 0ll;
            } else {
                t = SweepTrack{globalLog, 0, b >= 0 && b / 64 == group ? (long long)(1ull << (b & 63)) : 0ll};
            }
        }
    };
    setTracks(0);
    // a genome has a track of its own when something in scope hangs under it
    std::vector<char> hasTrack((size_t)ng, 0);
    for (int g : postOrder) {
        const GenomeTables &G = img.genomes[(size_t)g];
        if (G.numBot <= 0)
            continue;
        for (int c : G.children)
            if (inScope[(size_t)c] && img.genomes[(size_t)c].totalLength > 0 && img.genomes[(size_t)c].numTop > 0)
                hasTrack[(size_t)g] = 1;
    }
    // the genome at the top of the scope keeps the sizes of its sets, not the sets (sweepTracks: topSizes) — when its children go in
    // one launch and the sets are counted once (one group)
    bool topSizes = false;
    if (!sum && groups == 1 && hasTrack[(size_t)path[0]] && !(getenv("HGX_SWEEP_TOP_SIZES") && getenv("HGX_SWEEP_TOP_SIZES")[0] == '0')) {
        size_t kids = 0;
        for (int c : img.genomes[(size_t)path[0]].children)
            if (inScope[(size_t)c] && img.genomes[(size_t)c].totalLength > 0 && img.genomes[(size_t)c].numTop > 0)
                ++kids;
        topSizes = kids <= (size_t)SWEEP_MAX_CHILDREN;
    }
    size_t need = 0;
    for (int g : postOrder)
        if (hasTrack[(size_t)g] && !(topSizes && g == path[0]))
            need += (size_t)img.genomes[(size_t)g].totalLength << track[(size_t)g].wlog;
    // (a depth along the path is a byte when genome sets are counted: at most 64 of them a group)
    const size_t depthBytes = sum ? 4 : 1;
    for (size_t i = path.size() == 1 || topSizes ? 0 : 1; i < path.size(); ++i)
        need += (size_t)img.genomes[(size_t)path[i]].totalLength * depthBytes;
    size_t freeB = 0, totalB = 0;
    HIP_OK(hipMemGetInfo(&freeB, &totalB));
    if (need + (1ull << 30) > freeB)
        return false;
    std::vector<Buf> S((size_t)ng), A((size_t)ng);
    for (int g : postOrder)
        if (hasTrack[(size_t)g] && !(topSizes && g == path[0]))
            S[(size_t)g].resize((size_t)img.genomes[(size_t)g].totalLength << track[(size_t)g].wlog);
    for (size_t i = path.size() == 1 || topSizes ? 0 : 1; i < path.size(); ++i)
        A[(size_t)path[i]].resize((size_t)img.genomes[(size_t)path[i]].totalLength * depthBytes);
    Ev a, b;
    HIP_OK(hipEventRecord(a.e, s));
    for (int group = 0; group < groups; ++group) {
    if (group > 0) // this group's genomes carry their bit, the others none
        setTracks(group);
    if (h->dev->wide) {
        if (sum)
            sweepTracks<int64_t, true, int32_t>(h, postOrder, path, inScope, track, hasTrack, S, A, s);
        else
            sweepTracks<int64_t, false, uint8_t>(h, postOrder, path, inScope, track, hasTrack, S, A, s, false, topSizes);
    } else {
        if (sum)
            sweepTracks<int32_t, true, int32_t>(h, postOrder, path, inScope, track, hasTrack, S, A, s);
        else
            sweepTracks<int32_t, false, uint8_t>(h, postOrder, path, inScope, track, hasTrack, S, A, s, false, topSizes);
    }
    if (sum)
        hipLaunchKernelGGL(k_sweep_out<int32_t>, dim3(2048), dim3(256), 0, s, (const int32_t *)A[(size_t)ref].p, first, count, step,
                           mode == 2 || group > 0 ? 0 : 1, d_out, group > 0 ? 1 : 0);
    else if (step == 1 && group == 0 && count >= 4096)
        hipLaunchKernelGGL(k_sweep_out_bytes, dim3(std::min(4096, sweepGrid())), dim3(256), 0, s, (const uint8_t *)A[(size_t)ref].p, first, count, mode == 2 ? 0 : 1, d_out);
    else
        hipLaunchKernelGGL(k_sweep_out<uint8_t>, dim3(2048), dim3(256), 0, s, (const uint8_t *)A[(size_t)ref].p, first, count, step,
                           mode == 2 || group > 0 ? 0 : 1, d_out, group > 0 ? 1 : 0);
    }
    HIP_OK(hipEventRecord(b.e, s));
    HIP_OK(hipStreamSynchronize(s));
    if (stats) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a.e, b.e));
        stats->depth_ms += ms;
        stats->columns += (uint64_t)count;
        stats->sweep_bytes += (uint64_t)need;
    }
    return true;
}

void columnsDepthDevice(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                        int32_t *d_out, void *stream, ColumnStats *stats, bool countDerefs, bool perBase) {
    HIP_OK(hipSetDevice(h->dev->device));
    hipStream_t s = (hipStream_t)stream;
    if (!perBase && !perBaseColumns() && !countDerefs && columnsDepthSweep(h, ref, first, count, step, mode, opt, d_out, s, stats))
        return;
    Buf err(4);
    HIP_OK(hipMemsetAsync(err.p, 0, 4, s));
    Buf masks;
    ColumnParams P = makeParams(h, ref, first, count, step, opt, (unsigned int *)err.p, masks);
    Buf der(16);
    if (stats && countDerefs) {
        HIP_OK(hipMemsetAsync(der.p, 0, 16, s));
        P.derefs = (unsigned long long *)der.p;
    }
    Ev a, b;
    Buf longRuns, longCount(8);
    HIP_OK(hipEventRecord(a.e, s));
    if (count > 0 && (perBase || perBaseColumns())) {
        // one walk per column: 64 neighbouring columns per wavefront walk in lockstep.  Used for the row counts of the MAF
        // path (short chunks: one lane per reference SEGMENT would leave the GPU empty) and as a cross-check of the run
        // kernel (HGX_COLUMNS_PER_BASE=1)
        const int grid = (int)std::min<int64_t>(COL_GRID, (count + 255) / 256);
        if (h->dev->wide)
            LAUNCH_DEPTH(k_column_depth, int64_t, P, mode, d_out);
        else
            LAUNCH_DEPTH(k_column_depth, int32_t, P, mode, d_out);
    } else if (count > 0) {
        // one walk per run, one lane per reference segment
        int32_t segFirst;
        int64_t segCount;
        refSegmentRange(h, ref, first, first + (count - 1) * step, segFirst, segCount);
        const unsigned long long longCap = (unsigned long long)(count / LANE_FILL_MAX + 1);
        longRuns.resize((size_t)longCap * sizeof(LongRun));
        HIP_OK(hipMemsetAsync(longCount.p, 0, 8, s));
        const int grid = (int)std::min<int64_t>(COL_GRID, (segCount + 255) / 256);
        if (h->dev->wide)
            LAUNCH_DEPTH(k_depth_runs, int64_t, P, mode, segFirst, segCount, d_out, (LongRun *)longRuns.p, (unsigned long long *)longCount.p,
                         longCap);
        else
            LAUNCH_DEPTH(k_depth_runs, int32_t, P, mode, segFirst, segCount, d_out, (LongRun *)longRuns.p, (unsigned long long *)longCount.p,
                         longCap);
        hipLaunchKernelGGL(k_fill_long_runs, dim3(1024), dim3(256), 0, s, (const LongRun *)longRuns.p, (const unsigned long long *)longCount.p,
                           d_out);
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
        if (countDerefs) {
            unsigned long long d[2] = {0, 0};
            HIP_OK(hipMemcpy(d, der.p, 16, hipMemcpyDeviceToHost));
            stats->top_derefs += d[0];
            stats->bottom_derefs += d[1];
        }
    }
}

void columnsDepthHost(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                      int32_t *out, ColumnStats *stats, bool countDerefs) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    Buf d((size_t)count * 4);
    columnsDepthDevice(h, ref, first, count, step, mode, opt, (int32_t *)d.p, nullptr, stats, countDerefs);
    if (count > 0)
        HIP_OK(hipMemcpy(out, d.p, (size_t)count * 4, hipMemcpyDeviceToHost));
}

void columnsDepthChunksHost(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                            ColumnStats *stats, int64_t chunk, const std::function<void(const int32_t *, int64_t, int64_t)> &sink) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    if (count <= 0)
        return;
    if (chunk < 1)
        chunk = 1;
    Buf d((size_t)count * 4);
    columnsDepthDevice(h, ref, first, count, step, mode, opt, (int32_t *)d.p, nullptr, stats, false);
    // two page-locked blocks in turn: the copy of a chunk goes on while the chunk before is with the sink
    struct Block {
        int32_t *p = nullptr;
        ~Block() { hostBlockGive(p); }
    } block[2];
    const size_t bytes = (size_t)std::min(chunk, count) * 4;
    block[0].p = static_cast<int32_t *>(hostBlockTake(bytes));
    if (count > chunk)
        block[1].p = static_cast<int32_t *>(hostBlockTake(bytes));
    int32_t *const buffer[2] = {block[0].p, block[1].p};
    const int32_t *values = (const int32_t *)d.p;
    handOffChunks(buffer, count, chunk,
                  [values](int32_t *p, int64_t lo, int64_t n) { HIP_OK(hipMemcpy(p, values + lo, (size_t)n * 4, hipMemcpyDeviceToHost)); },
                  sink);
}

// halAlignmentDepth's lines made on the device (k_wig_text): the values never leave it; the text of a chunk of values comes to a
// page-locked block while the next chunk's is made, and goes to `sink` in order.
void columnsDepthTextChunksHost(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int mode, const ColumnOptions &opt,
                                ColumnStats *stats, int64_t chunk, const std::function<void(const char *, size_t)> &sink) {
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    if (count <= 0)
        return;
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, (int64_t)1 << 26)); // (a chunk's lines: 12 bytes each at most, offsets below 2^38)
    Buf d((size_t)count * 4);
    columnsDepthDevice(h, ref, first, count, step, mode, opt, (int32_t *)d.p, nullptr, stats, false);
    HIP_OK(hipDeviceSynchronize());
    const int32_t *values = (const int32_t *)d.p;
    struct Side {
        hipStream_t s = nullptr;
        Buf text, ctl; // ctl: WigCtl (16 bytes), then the scan's tiles
        int64_t lo = 0, n = 0;
        ~Side() {
            if (s) {
                (void)hipStreamSynchronize(s);
                (void)hipStreamDestroy(s);
            }
        }
    } side[2];
    const size_t tiles = (size_t)((std::min(chunk, count) + WIG_TILE - 1) / WIG_TILE) + 1;
    const int64_t numChunks = (count + chunk - 1) / chunk;
    for (int k = 0; k < 2 && k < numChunks; ++k) {
        HIP_OK(hipStreamCreateWithFlags(&side[k].s, hipStreamNonBlocking));
        side[k].text.resize((size_t)std::min(chunk, count) * 12);
        side[k].ctl.resize(16 + tiles * 8);
    }
    auto launch = [&](int64_t c) {
        Side &S = side[c & 1];
        S.lo = c * chunk;
        S.n = std::min(chunk, count - S.lo);
        HIP_OK(hipMemsetAsync(S.ctl.p, 0, 16 + tiles * 8, S.s));
        const uint32_t nt = (uint32_t)((S.n + WIG_TILE - 1) / WIG_TILE);
        hipLaunchKernelGGL(k_wig_text, dim3(nt), dim3(256), 0, S.s, values + S.lo, (uint32_t)S.n, (WigCtl *)S.ctl.p, (unsigned long long *)((char *)S.ctl.p + 16),
                           (char *)S.text.p);
    };
    auto finish = [&](int64_t c) {
        Side &S = side[c & 1];
        WigCtl ctl{0, 0, 0};
        HIP_OK(hipMemcpyAsync(&ctl, S.ctl.p, sizeof ctl, hipMemcpyDeviceToHost, S.s));
        HIP_OK(hipStreamSynchronize(S.s));
        const size_t bytes = (size_t)ctl.bytes;
        struct Block {
            char *p;
            ~Block() { hostBlockGive(p); }
        } block{static_cast<char *>(hostBlockTake(std::max<size_t>(bytes, 1)))};
        HIP_OK(hipMemcpyAsync(block.p, S.text.p, bytes, hipMemcpyDeviceToHost, S.s));
        HIP_OK(hipStreamSynchronize(S.s));
        sink(block.p, bytes);
    };
    for (int64_t c = 0; c < numChunks; ++c) {
        launch(c);
        if (c > 0)
            finish(c - 1);
    }
    finish(numChunks - 1);
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
    Buf masks;
    ColumnParams P = makeParams(h, ref, first, count, 1, opt, (unsigned int *)err.p, masks);
    Ev a, b;
    HIP_OK(hipEventRecord(a.e, nullptr));
    const int grid = (int)std::min<int64_t>(COL_GRID, (count + 255) / 256);
    if (h->dev->wide)
        hipLaunchKernelGGL((k_column_rows<int64_t, uint64_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint64_t *)dOff.p,
                           (ColumnRow *)dRows.p, (const uint8_t *)nullptr);
    else
        hipLaunchKernelGGL((k_column_rows<int32_t, uint64_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint64_t *)dOff.p,
                           (ColumnRow *)dRows.p, (const uint8_t *)nullptr);
    HIP_OK(hipEventRecord(b.e, nullptr));
    unsigned int e = 0;
    HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
    if (e)
        throw std::runtime_error("column walk exceeded the frame stack (more than 64 pending branches for one base, or a recursion deeper than 253)");
    if (total)
        HIP_OK(hipMemcpy(rows.data(), dRows.p, total * sizeof(ColumnRow), hipMemcpyDeviceToHost));
    if (stats) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a.e, b.e));
        stats->rows_ms += ms;
        stats->rows += total;
    }
}

void columnsGapRowsHost(hgx_alignment *h, int ref, const std::vector<GapAskHost> &asks, const ColumnOptions &opt, bool withDna,
                        std::vector<uint64_t> &rowOffset, std::vector<ColumnRowHost> &rows, ColumnStats *stats, bool events) {
    static_assert(sizeof(GapAskHost) == sizeof(GapAsk), "ask layouts must match");
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    HIP_OK(hipSetDevice(h->dev->device));
    if (withDna)
        ensureDeviceDna(h->img, *h->dev);
    const size_t n = asks.size();
    rowOffset.assign(n + 1, 0);
    rows.clear();
    if (n == 0)
        return;
    for (const GapAskHost &a : asks)
        if (a.genome < 0 || a.genome >= (int)h->img.genomes.size() || a.pos < 0 || a.pos >= h->img.genomes[(size_t)a.genome].totalLength)
            throw std::runtime_error("column asked for outside its genome");
    Buf dAsk(n * sizeof(GapAsk)), dCnt(n * 4), dOff((n + 1) * 8), err(4), masks;
    HIP_OK(hipMemcpy(dAsk.p, asks.data(), n * sizeof(GapAsk), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(err.p, 0, 4));
    ColumnParams P = makeParams(h, ref, 0, 0, 1, opt, (unsigned int *)err.p, masks);
    P.count = (int64_t)n;
    P.noGapEvents = events ? 0 : 1;
    Ev a, b;
    HIP_OK(hipEventRecord(a.e, nullptr));
    const int grid = (int)std::min<int64_t>(COL_GRID, ((int64_t)n + 255) / 256);
    if (h->dev->wide)
        hipLaunchKernelGGL((k_gap_count<int64_t>), dim3(grid), dim3(256), 0, nullptr, P, (const GapAsk *)dAsk.p, (uint32_t *)dCnt.p);
    else
        hipLaunchKernelGGL((k_gap_count<int32_t>), dim3(grid), dim3(256), 0, nullptr, P, (const GapAsk *)dAsk.p, (uint32_t *)dCnt.p);
    std::vector<uint32_t> cnt(n);
    HIP_OK(hipMemcpy(cnt.data(), dCnt.p, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
        rowOffset[i + 1] = rowOffset[i] + cnt[i];
    const uint64_t total = rowOffset[n];
    rows.resize(total);
    Buf dRows(std::max<uint64_t>(total, 1) * sizeof(ColumnRow));
    HIP_OK(hipMemcpy(dOff.p, rowOffset.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    if (h->dev->wide)
        hipLaunchKernelGGL((k_gap_rows<int64_t>), dim3(grid), dim3(256), 0, nullptr, P, (const GapAsk *)dAsk.p, (const uint64_t *)dOff.p,
                           (ColumnRow *)dRows.p);
    else
        hipLaunchKernelGGL((k_gap_rows<int32_t>), dim3(grid), dim3(256), 0, nullptr, P, (const GapAsk *)dAsk.p, (const uint64_t *)dOff.p,
                           (ColumnRow *)dRows.p);
    HIP_OK(hipEventRecord(b.e, nullptr));
    unsigned int e = 0;
    HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
    if (e)
        throw std::runtime_error("column walk exceeded the frame stack (more than 64 pending branches for one base, or a recursion deeper than 253)");
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
// 64-bit total of a uint32 array (the scan below wraps at 2^32: callers check the chunk before they trust its offsets)
static __global__ void __launch_bounds__(256) k_sum64(const uint32_t *__restrict__ in, uint32_t n, unsigned long long *total) {
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        s += in[i];
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s)
        atomicAdd(total, s);
}
static void checkRowTotal(const uint32_t *counts, uint32_t n) {
    Buf t(8);
    HIP_OK(hipMemset(t.p, 0, 8));
    hipLaunchKernelGGL(k_sum64, dim3(256), dim3(256), 0, nullptr, counts, n, (unsigned long long *)t.p);
    unsigned long long total = 0;
    HIP_OK(hipMemcpy(&total, t.p, 8, hipMemcpyDeviceToHost));
    if (total >= (1ull << 32))
        throw ColumnChunkTooLarge();
}

static uint32_t deviceScan(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *blockSums) {
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, nullptr, in, n, blockSums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, nullptr, blockSums, nb, out + n);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, nullptr, in, n, (const uint32_t *)blockSums, out);
    uint32_t total = 0;
    HIP_OK(hipMemcpy(&total, out + n, 4, hipMemcpyDeviceToHost));
    return total;
}

// ---- page-locked host blocks for the large copies (hgx_columns_engine.hpp) ----
namespace {
struct HostBlockHeader { // in front of every block
    size_t capacity; // bytes behind the header
    uint32_t pinned; // hipHostMalloc'ed (else malloc'ed)
    uint32_t magic;
    char pad[48];
};
static_assert(sizeof(HostBlockHeader) == 64, "blocks stay 64-byte aligned");
const uint32_t HOST_BLOCK_MAGIC = 0x48474258u;
// (locking pages costs about as much as copying them once through pageable memory: it pays for blocks that are used again, and a
// block of a genome's length is not)
const size_t HOST_BLOCK_SMALL = (size_t)1 << 20, HOST_BLOCK_PIN_MAX = (size_t)256 << 20, HOST_BLOCK_KEEP = 12,
             HOST_BLOCK_KEEP_BYTES = (size_t)1 << 30;
struct HostBlockPool {
    std::mutex mu;
    std::vector<HostBlockHeader *> idle;
    size_t idleBytes = 0;
    bool noDevice = false; // (hipHostMalloc said so once: not asked again)
};
HostBlockPool &hostBlockPool() {
    static HostBlockPool *pool = new HostBlockPool; // (never destroyed: blocks may be given back while the process ends)
    return *pool;
}
} // namespace

void *hostBlockTake(size_t bytes) {
    if (bytes == 0)
        bytes = 1;
    HostBlockHeader *hd = nullptr;
    if (bytes >= HOST_BLOCK_SMALL) {
        HostBlockPool &pool = hostBlockPool();
        bool tryPinned;
        {
            std::lock_guard<std::mutex> lock(pool.mu);
            size_t best = pool.idle.size();
            for (size_t i = 0; i < pool.idle.size(); ++i) // the smallest kept block that holds it (and is not four times too large)
                if (pool.idle[i]->capacity >= bytes && pool.idle[i]->capacity / 4 <= bytes &&
                    (best == pool.idle.size() || pool.idle[i]->capacity < pool.idle[best]->capacity))
                    best = i;
            if (best != pool.idle.size()) {
                hd = pool.idle[best];
                pool.idle.erase(pool.idle.begin() + (std::ptrdiff_t)best);
                pool.idleBytes -= hd->capacity;
                return hd + 1;
            }
            tryPinned = !pool.noDevice && bytes <= HOST_BLOCK_PIN_MAX;
        }
        const size_t capacity = (bytes + bytes / 8 + ((size_t)4 << 20) - 1) & ~(((size_t)4 << 20) - 1); // (an eighth of slack, 4 MiB steps)
        if (tryPinned) {
            void *p = nullptr;
            const hipError_t e = hipHostMalloc(&p, capacity + sizeof(HostBlockHeader), hipHostMallocPortable);
            if (e == hipSuccess && p) {
                hd = static_cast<HostBlockHeader *>(p);
                hd->capacity = capacity;
                hd->pinned = 1;
                hd->magic = HOST_BLOCK_MAGIC;
                return hd + 1;
            }
            (void)hipGetLastError();
            if (e == hipErrorNoDevice || e == hipErrorInsufficientDriver || e == hipErrorNotInitialized || e == hipErrorInvalidDevice) {
                std::lock_guard<std::mutex> lock(pool.mu);
                pool.noDevice = true;
            }
        }
        hd = static_cast<HostBlockHeader *>(aligned_alloc(64, (capacity + sizeof(HostBlockHeader) + 63) & ~(size_t)63));
        if (!hd)
            throw std::bad_alloc();
        hd->capacity = capacity;
        hd->pinned = 0;
        hd->magic = HOST_BLOCK_MAGIC;
        return hd + 1;
    }
    hd = static_cast<HostBlockHeader *>(aligned_alloc(64, (bytes + sizeof(HostBlockHeader) + 63) & ~(size_t)63));
    if (!hd)
        throw std::bad_alloc();
    hd->capacity = bytes;
    hd->pinned = 0;
    hd->magic = HOST_BLOCK_MAGIC;
    return hd + 1;
}

void hostBlockGive(void *p) noexcept {
    if (!p)
        return;
    HostBlockHeader *hd = static_cast<HostBlockHeader *>(p) - 1;
    if (hd->magic != HOST_BLOCK_MAGIC)
        std::abort(); // (not one of ours: nothing sensible to do)
    if (hd->capacity >= HOST_BLOCK_SMALL) {
        HostBlockPool &pool = hostBlockPool();
        std::lock_guard<std::mutex> lock(pool.mu);
        if (pool.idle.size() < HOST_BLOCK_KEEP && pool.idleBytes + hd->capacity <= HOST_BLOCK_KEEP_BYTES) {
            pool.idle.push_back(hd);
            pool.idleBytes += hd->capacity;
            return;
        }
    }
    hd->magic = 0;
    if (hd->pinned)
        (void)hipHostFree(hd);
    else
        free(hd);
}

// ---- hal2maf's device stage from piece closures (hgx_maf_kernels.hpp) ----
// The per-base tracks of one (reference, scope, filters): S (the reported bases in the tree below every base: the depth sweeps with
// sums), A of the reference (= the rows of every column), D / F (where a segment boundary lies between two neighbouring bases,
// below a base / anywhere in its column).  Built once by sweeps over whole genomes, kept with the handle: an export's chunks, and
// the slices of hgx_maf_export_multi, find them there.
namespace {
struct MafTracks {
    enum : int { UNCHECKED = 0, CHECKED = 1, REFUSED = 2 };
    std::atomic<int> state{UNCHECKED}; // the first chunk taken from the tracks is held against the column walk (columnsHeadRowsHost)
    std::atomic<int> stateUnique{UNCHECKED}; // ... the first chunk of an export with --unique (the stretches of hgx_maf_kernels.hpp)
    std::atomic<uint64_t> chunksUnique{0};
    std::atomic<uint64_t> chunks{0};   // chunks served
    std::atomic<uint64_t> deviceUs{0}, servedColumns{0}, servedHeads{0}, servedMarked{0}; // ... their kernels' time (HIP events), columns, heads
    int ref = -1;
    bool noAncestors = false, noDupes = false;
    std::vector<int> targets;
    int device = -1;
    std::vector<Buf> S, D, A, F;
    Buf sPtrs;               // device: const int32_t *[genomes]
    Buf refLocate;           // coarse position -> segment table of the reference's tiling (k_maf_locate_table)
    int refLocateShift = 0;
    const uint8_t *Fref = nullptr;
    const int32_t *Aref = nullptr; // null: every column has constRows rows
    int32_t constRows = 0;
    size_t bytes = 0;
    double buildMs = 0;
};
} // namespace

static bool mafSweepAllowed(const Image &img, const SweepScope &sc, int64_t exportColumns) {
    const char *env = getenv("HGX_MAF_SWEEP");
    if (env && env[0] == '0')
        return false;
    if (env && env[0] == '1')
        return true;
    // Measured on the MI355X (profiles/r06_notes.md, config 2's alignment): the sweeps that build the tracks touch every base of every
    // genome in scope once — 6.4 ms for a gigabase of scope; a batch of a million columns then costs 0.8 ms of kernels (k_maf_mark_list,
    // k_maf_rows_ctl, k_maf_heads_out, k_maf_ship) where the walk's launches cost 2.8 (k_column_depth, k_column_rows twice over every
    // column, k_column_heads, k_gather_head_rows): the tracks pay from about 300 bases of scope a column down.  The rule keeps a
    // threefold margin on that and leaves exports below a million columns to the walk (a first batch held against the walk, the
    // buffers of a stream: fixed costs of a few milliseconds).
    int64_t bases = 0;
    for (int g : sc.postOrder)
        bases += img.genomes[(size_t)g].totalLength;
    return exportColumns >= ((int64_t)1 << 20) && bases <= 100 * exportColumns;
}

template <typename C>
static void mafBreakSweeps(hgx_alignment *h, const SweepScope &sc, const std::vector<char> &hasTrack, MafTracks &M, hipStream_t s) {
    const Image &img = h->img;
    const DeviceImage &D = *h->dev;
    const int GRID = 4096;
    for (int g : sc.postOrder) {
        const GenomeTables &G = img.genomes[(size_t)g];
        if (!hasTrack[(size_t)g])
            continue;
        const DeviceGenome &dg = D.genomes[(size_t)g];
        std::vector<BreakChild> kids;
        for (size_t k = 0; k < G.children.size(); ++k) {
            const int c = G.children[k];
            if (sc.inScope[(size_t)c] && hasTrack[(size_t)c]) // (a child without bottom segments in scope adds no boundary of its own)
                kids.push_back(BreakChild{dg.childEnc[k], D.genomes[(size_t)c].top, (const uint8_t *)M.D[(size_t)c].p});
        }
        size_t at = 0;
        do {
            BreakChildren ch;
            ch.noRing = M.noDupes ? 1 : 0;
            ch.n = (int)std::min<size_t>(SWEEP_MAX_CHILDREN, kids.size() - at);
            for (int k = 0; k < ch.n; ++k)
                ch.c[k] = kids[at + (size_t)k];
            hipLaunchKernelGGL((k_break_up<C>), dim3(GRID), dim3(256), 0, s, (const BotRec<C> *)dg.bot, (int64_t)G.numBot, ch, at ? 1 : 0,
                               (uint8_t *)M.D[(size_t)g].p);
            at += SWEEP_MAX_CHILDREN;
        } while (at < kids.size());
    }
    const int top = sc.path[0];
    {
        const GenomeTables &G = img.genomes[(size_t)top];
        hipLaunchKernelGGL(k_break_top, dim3(GRID), dim3(256), 0, s, hasTrack[(size_t)top] ? (const uint8_t *)M.D[(size_t)top].p : (const uint8_t *)nullptr,
                           (int64_t)G.totalLength, (uint8_t *)M.F[(size_t)top].p);
        if (G.numTop > 0)
            hipLaunchKernelGGL((k_break_top_starts<C>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)D.genomes[(size_t)top].top, (int64_t)G.numTop,
                               (uint8_t *)M.F[(size_t)top].p);
    }
    for (size_t i = 1; i < sc.path.size(); ++i) {
        const int c = sc.path[i], p = sc.path[i - 1];
        const int32_t *pEnc = nullptr; // (--noDupes: the parent's links to the path's genome)
        if (M.noDupes) {
            const GenomeTables &P = img.genomes[(size_t)p];
            for (size_t k = 0; k < P.children.size(); ++k)
                if (P.children[k] == c)
                    pEnc = D.genomes[(size_t)p].childEnc[k];
        }
        hipLaunchKernelGGL((k_break_down<C>), dim3(GRID), dim3(256), 0, s, (const TopRec<C> *)D.genomes[(size_t)c].top, (int64_t)img.genomes[(size_t)c].numTop,
                           (const BotRec<C> *)D.genomes[(size_t)p].bot, (const uint8_t *)M.F[(size_t)p].p,
                           hasTrack[(size_t)c] ? (const uint8_t *)M.D[(size_t)c].p : (const uint8_t *)nullptr, (uint8_t *)M.F[(size_t)c].p, pEnc);
    }
}

// the tracks of (ref, opt) — from the handle, or built now; null when the sweeps do not pay or do not fit
static std::shared_ptr<MafTracks> mafTracksFor(hgx_alignment *h, int ref, const ColumnOptions &opt, int64_t exportColumns) {
    if (opt.onlyOrthologs)
        return nullptr; // (the paralogs of the path's own segments left out, the children's kept: the walk)
    // (--noDupes cuts the same edges everywhere — no ring is followed, only the segment a parent's slot names goes up: the sweeps and the
    // row selection take it as a flag)
    const Image &img = h->img;
    const int ng = (int)img.genomes.size();
    std::vector<int> tg(opt.targets.begin(), opt.targets.end());
    std::sort(tg.begin(), tg.end());
    tg.erase(std::unique(tg.begin(), tg.end()), tg.end());
    std::lock_guard<std::mutex> lock(h->mafTracksMutex);
    if (h->mafTracks) {
        std::shared_ptr<MafTracks> have = std::static_pointer_cast<MafTracks>(h->mafTracks);
        if (have->ref == ref && have->noAncestors == opt.noAncestors && have->noDupes == opt.noDupes && have->targets == tg && have->device == h->dev->device) {
            const char *env = getenv("HGX_MAF_SWEEP");
            return env && env[0] == '0' ? nullptr : have;
        }
    }
    const SweepScope sc = sweepScope(img, ref, opt);
    if (!mafSweepAllowed(img, sc, exportColumns))
        return nullptr;
    // a genome has tracks of its own when something in scope hangs under it
    std::vector<char> hasTrack((size_t)ng, 0);
    for (int g : sc.postOrder) {
        const GenomeTables &G = img.genomes[(size_t)g];
        if (G.numBot <= 0)
            continue;
        for (int c : G.children)
            if (sc.inScope[(size_t)c] && img.genomes[(size_t)c].totalLength > 0 && img.genomes[(size_t)c].numTop > 0)
                hasTrack[(size_t)g] = 1;
    }
    std::vector<SweepTrack> track((size_t)ng);
    for (int g = 0; g < ng; ++g)
        track[(size_t)g] = SweepTrack{2, 0, sc.counted[(size_t)g] ? 1ll : 0ll};
    const size_t PAD = 8;
    size_t need = 0;
    for (int g : sc.postOrder)
        if (hasTrack[(size_t)g])
            need += (size_t)img.genomes[(size_t)g].totalLength * 5 + PAD;
    const size_t aFrom = sc.path.size() == 1 ? 0 : 1;
    for (size_t i = 0; i < sc.path.size(); ++i)
        need += (size_t)img.genomes[(size_t)sc.path[i]].totalLength * (i >= aFrom ? 5 : 1) + PAD;
    size_t freeB = 0, totalB = 0;
    HIP_OK(hipMemGetInfo(&freeB, &totalB));
    // (one set of tracks a handle: the old one's memory counts as free — and is given up only when the new set fits: ADVICE r05)
    const size_t oldBytes = h->mafTracks ? std::static_pointer_cast<MafTracks>(h->mafTracks)->bytes : 0;
    if (need + (1ull << 30) > freeB + oldBytes)
        return nullptr;
    h->mafTracks.reset();
    std::shared_ptr<MafTracks> M(new MafTracks);
    M->ref = ref;
    M->noAncestors = opt.noAncestors;
    M->noDupes = opt.noDupes;
    M->targets = tg;
    M->device = h->dev->device;
    M->bytes = need;
    M->S = std::vector<Buf>((size_t)ng);
    M->D = std::vector<Buf>((size_t)ng);
    M->A = std::vector<Buf>((size_t)ng);
    M->F = std::vector<Buf>((size_t)ng);
    hipStream_t s = nullptr;
    for (int g : sc.postOrder)
        if (hasTrack[(size_t)g]) {
            const size_t len = (size_t)img.genomes[(size_t)g].totalLength;
            M->S[(size_t)g].resize(len * 4);
            M->D[(size_t)g].resize(len + PAD);
            HIP_OK(hipMemsetAsync(M->D[(size_t)g].p, 0, len + PAD, s));
        }
    for (size_t i = 0; i < sc.path.size(); ++i) {
        const size_t len = (size_t)img.genomes[(size_t)sc.path[i]].totalLength;
        if (i >= aFrom)
            M->A[(size_t)sc.path[i]].resize(std::max<size_t>(len, 1) * 4);
        M->F[(size_t)sc.path[i]].resize(len + PAD);
        HIP_OK(hipMemsetAsync(M->F[(size_t)sc.path[i]].p, 0, len + PAD, s));
    }
    Ev a, b;
    HIP_OK(hipEventRecord(a.e, s));
    if (h->dev->wide) {
        sweepTracks<int64_t, true, int32_t>(h, sc.postOrder, sc.path, sc.inScope, track, hasTrack, M->S, M->A, s, opt.noDupes);
        mafBreakSweeps<int64_t>(h, sc, hasTrack, *M, s);
    } else {
        sweepTracks<int32_t, true, int32_t>(h, sc.postOrder, sc.path, sc.inScope, track, hasTrack, M->S, M->A, s, opt.noDupes);
        mafBreakSweeps<int32_t>(h, sc, hasTrack, *M, s);
    }
    { // where a column's walk begins: the reference segment of its base, from a table of a bucket per four segments
        const GenomeTables &R = img.genomes[(size_t)ref];
        const int64_t nseg = R.numTop > 0 ? R.numTop : R.numBot;
        if (nseg > 0 && R.totalLength > 0) {
            int64_t buckets = 1;
            while (buckets < nseg / 4 && buckets < ((int64_t)1 << 22))
                buckets <<= 1;
            int shift = 0;
            while (((R.totalLength - 1) >> shift) >= buckets)
                ++shift;
            const int64_t nb = ((R.totalLength - 1) >> shift) + 1;
            M->refLocate.resize(((size_t)nb + 1) * 4);
            M->refLocateShift = shift;
            const DeviceGenome &dg = h->dev->genomes[(size_t)ref];
            const int grid = (int)std::min<int64_t>(4096, (nb + 256) / 256);
            if (R.numTop > 0) {
                if (h->dev->wide)
                    hipLaunchKernelGGL((k_maf_locate_table<TopRec<int64_t>>), dim3(grid), dim3(256), 0, s, (const TopRec<int64_t> *)dg.top, nseg, shift, (uint32_t)nb, (int32_t *)M->refLocate.p);
                else
                    hipLaunchKernelGGL((k_maf_locate_table<TopRec<int32_t>>), dim3(grid), dim3(256), 0, s, (const TopRec<int32_t> *)dg.top, nseg, shift, (uint32_t)nb, (int32_t *)M->refLocate.p);
            } else {
                if (h->dev->wide)
                    hipLaunchKernelGGL((k_maf_locate_table<BotRec<int64_t>>), dim3(grid), dim3(256), 0, s, (const BotRec<int64_t> *)dg.bot, nseg, shift, (uint32_t)nb, (int32_t *)M->refLocate.p);
                else
                    hipLaunchKernelGGL((k_maf_locate_table<BotRec<int32_t>>), dim3(grid), dim3(256), 0, s, (const BotRec<int32_t> *)dg.bot, nseg, shift, (uint32_t)nb, (int32_t *)M->refLocate.p);
            }
        }
    }
    HIP_OK(hipEventRecord(b.e, s));
    std::vector<const int32_t *> ptrs((size_t)ng, nullptr);
    for (int g = 0; g < ng; ++g)
        if (hasTrack[(size_t)g])
            ptrs[(size_t)g] = (const int32_t *)M->S[(size_t)g].p;
    M->sPtrs.resize((size_t)ng * sizeof(void *));
    HIP_OK(hipMemcpyAsync(M->sPtrs.p, ptrs.data(), (size_t)ng * sizeof(void *), hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, a.e, b.e));
    M->buildMs = ms;
    M->Fref = (const uint8_t *)M->F[(size_t)ref].p;
    // (the rows of a column: A of the reference — sweepTracks leaves it for every genome of the path below the top, and for the
    // top itself when the reference is the top)
    M->Aref = (const int32_t *)M->A[(size_t)ref].p;
    M->constRows = sc.counted[(size_t)ref] ? 1 : 0;
    if (!hasTrack[(size_t)ref] && sc.path.size() == 1)
        M->Aref = nullptr; // (nothing in scope but the reference itself: k_sweep_top wrote the constant; no need to read it)
    if (getenv("HGX_MAF_TIMING"))
        fprintf(stderr, "[hgx] hal2maf tracks of genome %d: %.3f ms on the device, %.1f MB\n", ref, ms, (double)need / 1e6);
    h->mafTracks = M;
    return M;
}

std::string mafTracksInfo(hgx_alignment *h) {
    std::lock_guard<std::mutex> lock(h->mafTracksMutex);
    if (!h->mafTracks)
        return "{\"tracks\": false}";
    const MafTracks &T = *std::static_pointer_cast<MafTracks>(h->mafTracks);
    char buf[768];
    static const char *const states[] = {"unchecked", "checked against the column walk", "refused: the column walk is used"};
    snprintf(buf, sizeof buf,
             "{\"tracks\": true, \"reference\": %d, \"no_ancestors\": %s, \"targets\": %zu, \"build_ms\": %.4f, \"bytes\": %zu, \"state\": \"%s\", "
             "\"chunks_served\": %llu, \"state_unique\": \"%s\", \"chunks_served_unique\": %llu, \"columns_served\": %llu, \"marked_columns\": %llu, "
             "\"heads\": %llu, \"device_ms_served\": %.3f}",
             T.ref, T.noAncestors ? "true" : "false", T.targets.size(), T.buildMs, T.bytes, states[T.state.load()], (unsigned long long)T.chunks.load(),
             states[T.stateUnique.load()], (unsigned long long)T.chunksUnique.load(),
             (unsigned long long)T.servedColumns.load(), (unsigned long long)T.servedMarked.load(), (unsigned long long)T.servedHeads.load(),
             (double)T.deviceUs.load() / 1e3);
    return buf;
}

void mafTracksDrop(hgx_alignment *h) {
    std::lock_guard<std::mutex> lock(h->mafTracksMutex);
    h->mafTracks.reset();
}

namespace {
struct MafSizesDoNotAddUp : std::runtime_error {
    MafSizesDoNotAddUp() : std::runtime_error("hal2maf: a column's rows do not add up to its subtrees' sizes") {}
};
struct MafUniqueNeedsTheWalk : std::runtime_error { // (this chunk only: a column with more reference copies than a lane holds)
    MafUniqueNeedsTheWalk() : std::runtime_error("hal2maf --unique: a column's reference bases cannot be told from its rows") {}
};
} // namespace

// columnsHeadRowsHost from the tracks: the marked columns of the chunk, their rows by rank, the heads among them
// uniqueFirst >= 0 (--unique): the classes of the columns (passed over / walked for their keys / written) by stretches of the marked
// columns' runs, in the format of the walk (marks 2 and 3)
static void columnsHeadRowsSweep(hgx_alignment *h, MafTracks &T, int ref, int64_t first, int64_t count, const ColumnOptions &opt,
                                 std::vector<uint8_t> &head, std::vector<uint32_t> &headOffset, HeadRows &headRows, ColumnStats *stats,
                                 int64_t uniqueFirst) {
    const uint32_t n = (uint32_t)count;
    const int GRID = 1024;
    Buf dMark((size_t)n * 4), dRowsOf((size_t)n * 4), dMarkIdx(((size_t)n + 1) * 4), dRowOff(((size_t)n + 1) * 4), dSums(((size_t)n / SCAN_BLOCK + 2) * 4),
        err(4), masks;
    Ev e0, e1;
    HIP_OK(hipEventRecord(e0.e, nullptr));
    HIP_OK(hipMemset(err.p, 0, 4));
    MafRowParams M;
    M.P = makeParams(h, ref, first, count, 1, opt, (unsigned int *)err.p, masks);
    hipLaunchKernelGGL(k_maf_marks, dim3(GRID), dim3(256), 0, nullptr, T.Fref, T.Aref, T.constRows, first, n, (uint32_t *)dMark.p, (uint32_t *)dRowsOf.p);
    checkRowTotal((const uint32_t *)dRowsOf.p, n);
    const uint32_t nCand = deviceScan((const uint32_t *)dMark.p, n, (uint32_t *)dMarkIdx.p, (uint32_t *)dSums.p);
    const uint32_t totalRows = deviceScan((const uint32_t *)dRowsOf.p, n, (uint32_t *)dRowOff.p, (uint32_t *)dSums.p);
    Buf dCandCol((size_t)nCand * 4), dCandRow(((size_t)nCand + 1) * 4), dRows(std::max<size_t>(totalRows, 1) * sizeof(ColumnRow));
    hipLaunchKernelGGL(k_maf_list, dim3(GRID), dim3(256), 0, nullptr, (const uint32_t *)dMark.p, (const uint32_t *)dMarkIdx.p, (const uint32_t *)dRowOff.p, n,
                       (uint32_t *)dCandCol.p, (uint32_t *)dCandRow.p);
    HIP_OK(hipMemcpyAsync((uint32_t *)dCandRow.p + nCand, (const uint32_t *)dRowOff.p + n, 4, hipMemcpyDeviceToDevice, nullptr));
    M.S = (const int32_t *const *)T.sPtrs.p;
    M.refLocate = (const int32_t *)T.refLocate.p;
    M.refLocateShift = T.refLocateShift;
    M.candCol = (const uint32_t *)dCandCol.p;
    M.candRow = (const uint32_t *)dCandRow.p;
    M.nCand = nCand;
    const int rowGrid = (int)std::min<int64_t>(COL_GRID, ((int64_t)nCand * (1 << MAF_LPC_LOG) + 255) / 256);
    if (h->dev->wide)
        hipLaunchKernelGGL((k_maf_rows<int64_t>), dim3(std::max(rowGrid, 1)), dim3(256), 0, nullptr, M, (ColumnRow *)dRows.p);
    else
        hipLaunchKernelGGL((k_maf_rows<int32_t>), dim3(std::max(rowGrid, 1)), dim3(256), 0, nullptr, M, (ColumnRow *)dRows.p);
    uint32_t nHeads = 0, totalHeadRows = 0;
    Buf dHead(n), dHeadOffset, dOut;
    // (The blocks of both branches below live to the end of the function — round 6, the last day.  They were the branches' own, and a
    // branch ends with a LAUNCH: its blocks went back to the device cache while k_unique_gather / k_maf_gather were still reading them,
    // and another thread — a slice of hgx_maf_export_multi beside this one, asking for blocks of exactly these sizes — got them and
    // wrote into them: wild offsets, a GPU memory access fault every few passes.  One thread alone never saw it: its own next use of such
    // a block is queued behind the kernel.  A block is released behind a blocking copy, never behind a launch.)
    Buf dSegCnt, dSegOff, dSeg, dUnits, dUnitRows, dUnitOff, dUnitRowOff, dSums2, dIsHead, dHeadCnt, dHeadIdx, dHeadRowOff;
    HIP_OK(hipMemsetAsync(dHead.p, 0, n, nullptr));
    if (uniqueFirst >= 0) {
        if (uniqueFirst > first)
            throw std::runtime_error("columnsHeadRowsHost: the range of --unique begins behind its columns");
        UniqueParams U;
        U.candCol = (const uint32_t *)dCandCol.p;
        U.candRow = (const uint32_t *)dCandRow.p;
        U.rows = (const ColumnRow *)dRows.p;
        U.nCand = nCand;
        U.n = n;
        U.first = first;
        U.f = uniqueFirst;
        U.ref = ref;
        U.maxRefRows = UNIQUE_MAX_REF_ROWS;
        if (const char *e = getenv("HGX_MAF_UNIQUE_MAX_REF"))
            U.maxRefRows = std::max(0, std::min(UNIQUE_MAX_REF_ROWS, atoi(e)));
        U.error = (unsigned int *)err.p;
        // (the chunk that is held against the column walk keeps the walk's form: every such column's rows)
        U.collapseKeysOnly = T.stateUnique.load() == MafTracks::CHECKED && !(getenv("HGX_MAF_UNIQUE_COLLAPSE") && getenv("HGX_MAF_UNIQUE_COLLAPSE")[0] == '0') ? 1 : 0;
        const int candGrid = (int)std::max<int64_t>(1, std::min<int64_t>(GRID, ((int64_t)nCand + 255) / 256));
        dSegCnt.resize((size_t)nCand * 4);
        dSegOff.resize(((size_t)nCand + 1) * 4);
        hipLaunchKernelGGL(k_unique_count, dim3(candGrid), dim3(256), 0, nullptr, U, (uint32_t *)dSegCnt.p);
        const uint32_t nSeg = deviceScan((const uint32_t *)dSegCnt.p, nCand, (uint32_t *)dSegOff.p, (uint32_t *)dSums.p);
        {
            unsigned int early = 0; // (a column whose reference rows could not be held has no stretches: nothing to go on with)
            HIP_OK(hipMemcpy(&early, err.p, 4, hipMemcpyDeviceToHost));
            if (early == 3 || nSeg == 0)
                throw MafUniqueNeedsTheWalk();
        }
        dSeg.resize(std::max<size_t>(nSeg, 1) * sizeof(UniqueSeg));
        dUnits.resize(std::max<size_t>(nSeg, 1) * 4);
        dUnitRows.resize(std::max<size_t>(nSeg, 1) * 4);
        dUnitOff.resize(((size_t)nSeg + 1) * 4);
        dUnitRowOff.resize(((size_t)nSeg + 1) * 4);
        dSums2.resize(((size_t)nSeg / SCAN_BLOCK + 2) * 4);
        hipLaunchKernelGGL(k_unique_stretches, dim3(candGrid), dim3(256), 0, nullptr, U, (const uint32_t *)dSegOff.p, (UniqueSeg *)dSeg.p);
        const int segGrid = (int)std::max<int64_t>(1, std::min<int64_t>(GRID, ((int64_t)nSeg + 255) / 256));
        hipLaunchKernelGGL(k_unique_units, dim3(segGrid), dim3(256), 0, nullptr, U, (const UniqueSeg *)dSeg.p, nSeg, (uint32_t *)dUnits.p,
                           (uint32_t *)dUnitRows.p);
        checkRowTotal((const uint32_t *)dUnitRows.p, nSeg);
        nHeads = deviceScan((const uint32_t *)dUnits.p, nSeg, (uint32_t *)dUnitOff.p, (uint32_t *)dSums2.p);
        totalHeadRows = deviceScan((const uint32_t *)dUnitRows.p, nSeg, (uint32_t *)dUnitRowOff.p, (uint32_t *)dSums2.p);
        dHeadOffset.resize(((size_t)nHeads + 1) * 4);
        dOut.resize(std::max<size_t>(totalHeadRows, 1) * sizeof(ColumnRow));
        hipLaunchKernelGGL(k_unique_gather, dim3(segGrid), dim3(256), 0, nullptr, U, M.P.desc, (const UniqueSeg *)dSeg.p, nSeg, (const uint32_t *)dUnits.p,
                           (const uint32_t *)dUnitOff.p, (const uint32_t *)dUnitRowOff.p, (uint8_t *)dHead.p, (uint32_t *)dHeadOffset.p,
                           (ColumnRow *)dOut.p);
    } else {
        dIsHead.resize((size_t)nCand * 4);
        dHeadCnt.resize((size_t)nCand * 4);
        dHeadIdx.resize(((size_t)nCand + 1) * 4);
        dHeadRowOff.resize(((size_t)nCand + 1) * 4);
        hipLaunchKernelGGL(k_maf_heads, dim3(GRID), dim3(256), 0, nullptr, (const uint32_t *)dCandCol.p, (const uint32_t *)dCandRow.p,
                           (const ColumnRow *)dRows.p, nCand, (uint32_t *)dIsHead.p, (uint32_t *)dHeadCnt.p);
        nHeads = deviceScan((const uint32_t *)dIsHead.p, nCand, (uint32_t *)dHeadIdx.p, (uint32_t *)dSums.p);
        totalHeadRows = deviceScan((const uint32_t *)dHeadCnt.p, nCand, (uint32_t *)dHeadRowOff.p, (uint32_t *)dSums.p);
        dHeadOffset.resize(((size_t)nHeads + 1) * 4);
        dOut.resize(std::max<size_t>(totalHeadRows, 1) * sizeof(ColumnRow));
        hipLaunchKernelGGL(k_maf_gather, dim3(std::max(rowGrid, 1)), dim3(256), 0, nullptr, (const uint32_t *)dCandCol.p, (const uint32_t *)dCandRow.p,
                           (const ColumnRow *)dRows.p, nCand, (const uint32_t *)dIsHead.p, (const uint32_t *)dHeadIdx.p, (const uint32_t *)dHeadRowOff.p,
                           (uint8_t *)dHead.p, (uint32_t *)dHeadOffset.p, (ColumnRow *)dOut.p);
    }
    HIP_OK(hipEventRecord(e1.e, nullptr));
    unsigned int e = 0;
    HIP_OK(hipMemcpy(&e, err.p, 4, hipMemcpyDeviceToHost));
    if (e == 2)
        throw MafSizesDoNotAddUp();
    if (e == 3)
        throw MafUniqueNeedsTheWalk();
    if (e)
        throw std::runtime_error("column walk exceeded the frame stack (more than 64 pending branches for one base)");
    head.resize(n);
    HIP_OK(hipMemcpy(head.data(), dHead.p, n, hipMemcpyDeviceToHost));
    headOffset.resize((size_t)nHeads + 1);
    if (nHeads)
        HIP_OK(hipMemcpy(headOffset.data(), dHeadOffset.p, (size_t)nHeads * 4, hipMemcpyDeviceToHost));
    headOffset[nHeads] = totalHeadRows;
    headRows.resize(totalHeadRows);
    if (totalHeadRows)
        HIP_OK(hipMemcpy(headRows.data(), dOut.p, (size_t)totalHeadRows * sizeof(ColumnRow), hipMemcpyDeviceToHost));
    {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0.e, e1.e));
        T.deviceUs.fetch_add((uint64_t)(ms * 1e3));
        T.servedColumns.fetch_add((uint64_t)count);
        T.servedHeads.fetch_add(nHeads);
        T.servedMarked.fetch_add(nCand);
        if (stats) {
            stats->rows_ms += ms;
            stats->rows += totalRows;
            stats->columns += (uint64_t)count;
        }
    }
}

// columnsHeadRowsHost by the column walk: every column's rows on the device (two walks of every column: count, emit), the heads
// found by comparing neighbours, their rows gathered — the path of --noDupes and --onlyOrthologs, of exports too short for the sweeps
// to pay, and what the sweeps' first chunk is held against
static void columnsHeadRowsWalk(hgx_alignment *h, int ref, int64_t first, int64_t count, const ColumnOptions &opt, std::vector<uint8_t> &head,
                                std::vector<uint32_t> &headOffset, HeadRows &headRows, ColumnStats *stats, int64_t uniqueFirst);

// One chunk's heads at a time (round 6, the last day).  This path — the export of one batch, the first chunk's check — puts its
// launches, its scans' blocking four-byte copies and its results' copies on the NULL stream and was written for one thread;
// hgx_maf_export_multi runs several slices at a time.  With four or six of them in here side by side about every eighth pass over
// config 3 ended in a GPU memory access fault or a hang; with the path one at a time none did, and the passes were no slower (the
// threads no longer wait for each other's blocking copies).  The cause, found a few hours later (profiles/r06_notes.md 12):
// columnsHeadRowsSweep released the blocks of its two branches behind a LAUNCH instead of behind a blocking copy, and a neighbour
// asking for blocks of the same sizes got them while the kernel was still reading — mended there.  The lock stays: it costs nothing,
// and the path's other assumptions of being alone on the null stream have only ever been tested with it.  The batches of a large
// export go through MafChunkStream, streams of their own, and are not held up here.
namespace {
struct HeadPathOnly {
    static std::mutex &mu() {
        static std::mutex *m = new std::mutex;
        return *m;
    }
    std::unique_lock<std::mutex> lock;
    HeadPathOnly() : lock(mu(), std::defer_lock) {
        if (!(getenv("HGX_MAF_HEADS_LOCK") && getenv("HGX_MAF_HEADS_LOCK")[0] == '0'))
            lock.lock();
    }
};
} // namespace

void columnsHeadRowsHost(hgx_alignment *h, int ref, int64_t first, int64_t count, const ColumnOptions &opt, bool withDna,
                         std::vector<uint8_t> &head, std::vector<uint32_t> &headOffset, HeadRows &headRows, ColumnStats *stats,
                         int64_t uniqueFirst, int64_t exportColumns) {
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
    // the heads from the per-base tracks (hgx_maf_kernels.hpp) where the sweeps behind them pay
    if (std::shared_ptr<MafTracks> T = mafTracksFor(h, ref, opt, std::max(exportColumns, count))) {
        const char *env = getenv("HGX_MAF_SWEEP");
        const bool forced = env && env[0] == '1';
        const bool unique = uniqueFirst >= 0;
        std::atomic<int> &state = unique ? T->stateUnique : T->state;
        // (--unique tells a column's reference bases from its rows: the reference has to be reported)
        if (state.load() != MafTracks::REFUSED && !(unique && T->constRows == 0)) {
            bool good = true, refuse = true;
            try {
                HeadPathOnly only;
                columnsHeadRowsSweep(h, *T, ref, first, count, opt, head, headOffset, headRows, stats, uniqueFirst);
            } catch (const MafSizesDoNotAddUp &) {
                if (forced)
                    throw; // (forced: the tests want to see it)
                good = false;
            } catch (const MafUniqueNeedsTheWalk &) {
                good = false;
                refuse = false; // (this chunk by the walk; the next one may do without)
            }
            // The first chunk taken from a set of tracks is held against the column walk (a chunk of the export is walked once
            // more: a fiftieth of config 3): tracks whose heads are not the walk's are not used again, and the walk's answer goes out.
            if (good && state.load() == MafTracks::UNCHECKED) {
                std::vector<uint8_t> head2;
                std::vector<uint32_t> off2(1, 0);
                HeadRows rows2;
                {
                    HeadPathOnly only;
                    columnsHeadRowsWalk(h, ref, first, count, opt, head2, off2, rows2, nullptr, uniqueFirst);
                }
                good = head2 == head && off2 == headOffset && rows2.size() == headRows.size() &&
                       (rows2.empty() || memcmp(rows2.data(), headRows.data(), rows2.size() * sizeof(ColumnRowHost)) == 0);
                if (good) {
                    state.store(MafTracks::CHECKED);
                } else {
                    if (forced)
                        throw std::runtime_error("hal2maf: the heads taken from the per-base tracks differ from the column walk's");
                    head.swap(head2);
                    headOffset.swap(off2);
                    headRows.swap(rows2);
                    state.store(MafTracks::REFUSED);
                    fprintf(stderr, "[hgx] hal2maf: the heads taken from the per-base tracks differ from the column walk's; the walk is used\n");
                    return;
                }
            }
            if (good) {
                (unique ? T->chunksUnique : T->chunks).fetch_add(1);
                return;
            }
            if (refuse)
                state.store(MafTracks::REFUSED);
            head.clear();
            headOffset.assign(1, 0);
            headRows.clear();
        }
    }
    HeadPathOnly only;
    columnsHeadRowsWalk(h, ref, first, count, opt, head, headOffset, headRows, stats, uniqueFirst);
}

static void columnsHeadRowsWalk(hgx_alignment *h, int ref, int64_t first, int64_t count, const ColumnOptions &opt, std::vector<uint8_t> &head,
                                std::vector<uint32_t> &headOffset, HeadRows &headRows, ColumnStats *stats, int64_t uniqueFirst) {
    head.clear();
    headOffset.assign(1, 0);
    headRows.clear();
    const uint32_t n = (uint32_t)count;
    Buf dCnt((size_t)n * 4), dOff(((size_t)n + 1) * 4), dSums(((size_t)n / SCAN_BLOCK + 2) * 4), err(4);
    Ev e0, e1;
    HIP_OK(hipEventRecord(e0.e, nullptr));
    // 1. rows per column (--unique: and which columns the iterator walks and writes), 2. offsets, 3. all rows (device only)
    HIP_OK(hipMemset(err.p, 0, 4));
    Buf masks;
    ColumnParams P = makeParams(h, ref, first, count, 1, opt, (unsigned int *)err.p, masks);
    const int grid = (int)std::min<int64_t>(COL_GRID, (count + 255) / 256);
    const bool unique = uniqueFirst >= 0;
    Buf dCls;
    if (unique) {
        if (uniqueFirst > first)
            throw std::runtime_error("columnsHeadRowsHost: the range of --unique begins behind its columns");
        dCls.resize(n);
        if (h->dev->wide)
            hipLaunchKernelGGL((k_column_unique_count<int64_t>), dim3(grid), dim3(256), 0, nullptr, P, uniqueFirst, (int32_t *)dCnt.p, (uint8_t *)dCls.p);
        else
            hipLaunchKernelGGL((k_column_unique_count<int32_t>), dim3(grid), dim3(256), 0, nullptr, P, uniqueFirst, (int32_t *)dCnt.p, (uint8_t *)dCls.p);
    } else {
        columnsDepthDevice(h, ref, first, count, 1, 2, opt, (int32_t *)dCnt.p, nullptr, nullptr, false, true);
    }
    const uint8_t *cls = unique ? (const uint8_t *)dCls.p : nullptr;
    checkRowTotal((const uint32_t *)dCnt.p, n);
    const uint32_t totalRows = deviceScan((const uint32_t *)dCnt.p, n, (uint32_t *)dOff.p, (uint32_t *)dSums.p);
    Buf dRows((size_t)totalRows * sizeof(ColumnRow));
    if (h->dev->wide)
        hipLaunchKernelGGL((k_column_rows<int64_t, uint32_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint32_t *)dOff.p,
                           (ColumnRow *)dRows.p, cls);
    else
        hipLaunchKernelGGL((k_column_rows<int32_t, uint32_t>), dim3(grid), dim3(256), 0, nullptr, P, (const uint32_t *)dOff.p,
                           (ColumnRow *)dRows.p, cls);
    // 4. run heads, 5. offsets of the heads' rows, 6. gather them
    Buf dHead(n), dHeadCnt((size_t)n * 4), dHeadOff(((size_t)n + 1) * 4);
    hipLaunchKernelGGL(k_column_heads, dim3(grid), dim3(256), 0, nullptr, (const uint32_t *)dOff.p, (const ColumnRow *)dRows.p, count,
                       (uint8_t *)dHead.p, (uint32_t *)dHeadCnt.p, cls);
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
    struct Counts { // (a page-locked block, written by the copy: no need to clear it first)
        uint32_t *p;
        explicit Counts(size_t n) : p(static_cast<uint32_t *>(hostBlockTake(n * 4))) {}
        ~Counts() { hostBlockGive(p); }
        uint32_t operator[](size_t i) const { return p[i]; }
    } headCnt(n);
    HIP_OK(hipMemcpy(headCnt.p, dHeadCnt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    headRows.resize(totalHeadRows);
    if (totalHeadRows)
        HIP_OK(hipMemcpy(headRows.data(), dOut.p, (size_t)totalHeadRows * sizeof(ColumnRow), hipMemcpyDeviceToHost));
    uint32_t acc = 0;
    const uint8_t *marks = head.data();
    for (uint32_t c = 0; c < n;) {
        if (c + 8 <= n) { // (one column in twenty-odd is a head: eight marks at a time)
            uint64_t w;
            memcpy(&w, marks + c, 8);
            if (!(w & 0x0101010101010101ull)) {
                c += 8;
                continue;
            }
        }
        if (marks[c] & 1) { // (1: a head, 3: a column of --unique that is walked but not written; both have their rows here)
            acc += headCnt[c];
            headOffset.push_back(acc);
        }
        ++c;
    }
    if (stats) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0.e, e1.e));
        stats->rows_ms += ms;
        stats->rows += totalRows;
        stats->columns += (uint64_t)count;
    }
}


// ---- the batches of an export as a stream (hgx_maf_kernels.hpp: a chunk in one pass) ----
// columnsHeadRowsSweep makes a chunk with seventeen launches, six counts read back in between and three blocking copies into
// pageable memory, and hands the host rows it has to describe and sort.  Here a chunk is three launches on a stream of its own — the
// marked columns listed by a one-pass scan, their rows, the heads among them picked, described and sorted and written straight into
// page-locked host memory — one copy of the head marks, and ONE wait for the host; the next chunk's launches are queued before this
// one is waited for.  Plain exports (no --unique: its stretches keep the launches above).
namespace {
// Host memory the stream's kernels write (heads' offsets, columns and rows): page-locked whatever its size — hostBlockTake hands out
// ordinary memory below a megabyte and when locking fails, which a copy can live with and a kernel's store cannot.  One allocation a
// stream, carved up; a released one is kept for the next stream (hgx_maf_export_multi opens one per slice).
struct PinnedArena {
    char *p = nullptr;
    size_t capacity = 0;
};
struct PinnedArenas {
    std::mutex mu;
    std::vector<PinnedArena> idle;
};
PinnedArenas &pinnedArenas() {
    static PinnedArenas *a = new PinnedArenas;
    return *a;
}
PinnedArena pinnedArenaTake(size_t bytes) { // p == nullptr: no page-locked memory to be had
    PinnedArenas &A = pinnedArenas();
    {
        std::lock_guard<std::mutex> lock(A.mu);
        for (size_t i = 0; i < A.idle.size(); ++i)
            if (A.idle[i].capacity >= bytes && A.idle[i].capacity / 4 <= bytes) {
                PinnedArena a = A.idle[i];
                A.idle.erase(A.idle.begin() + (std::ptrdiff_t)i);
                return a;
            }
    }
    PinnedArena a;
    a.capacity = (bytes + (bytes >> 3) + ((size_t)1 << 16) - 1) & ~(((size_t)1 << 16) - 1);
    void *p = nullptr;
    if (hipHostMalloc(&p, a.capacity, hipHostMallocPortable) != hipSuccess || !p) {
        (void)hipGetLastError();
        return PinnedArena();
    }
    a.p = static_cast<char *>(p);
    return a;
}
void pinnedArenaGive(PinnedArena a) {
    if (!a.p)
        return;
    PinnedArenas &A = pinnedArenas();
    {
        std::lock_guard<std::mutex> lock(A.mu);
        if (A.idle.size() < 3) {
            A.idle.push_back(a);
            return;
        }
    }
    (void)hipHostFree(a.p);
}
static void describeHostRow(const Image &img, const std::vector<int32_t> &rankBase, const ColumnRowHost &r, MafChunkRow &out, uint32_t ord) {
    const GenomeTables &G = img.genomes[(size_t)r.genome];
    const int s = G.seqs.size() == 1 ? 0 : G.seqIndexBySite(r.pos);
    const SeqInfo &S = G.seqs[(size_t)s];
    const int64_t at = r.pos - S.start;
    out.key = r.rev ? ((S.length - 1 - at) << 1) | 1 : at << 1;
    out.rank = rankBase[(size_t)r.genome] + s;
    out.ord = ord;
}
} // namespace

struct MafChunkStream {
    static const int SLOTS = 3; // (a stream each: the launches of consecutive batches are short of lanes and latency-bound — they run beside each other)
    hgx_alignment *h = nullptr;
    std::shared_ptr<MafTracks> T;
    int ref = 0;
    ColumnOptions opt;
    std::vector<int32_t> rankBase;
    bool forced = false;
    Buf masks, dRankBase;
    ColumnParams P;
    hipStream_t streams[SLOTS] = {nullptr, nullptr, nullptr};
    ColumnStats *stats = nullptr;
    uint32_t maxChunk = 0, headRoom = 0, segRoom = 0;
    uint64_t rowsRoom = 0, outRoom = 0;
    size_t tiles1 = 0, tiles2 = 0;
    int64_t uniqueFirst = -1; // >= 0: hal2maf --unique over a range that begins there (the stretches of hgx_maf_kernels.hpp)
    struct Slot {
        Buf candCol, candRow, rows, head, ctl; // ctl: MafChunkCtl (64 bytes), then the two scans' tiles
        Buf headOff, headCol, out;             // what k_maf_heads_out writes and k_maf_ship sends to the host
        Buf seg;                               // --unique: the batch's stretches
        bool collapsed = false;                // --unique: the stretches walked for their keys shipped as one column each
        struct Host {
            void *p = nullptr;
        } hHead, hHeadOff, hHeadCol, hOut, hCtl; // (pieces of the stream's page-locked arena)
        Ev e0, e1;
        int64_t first = 0, count = 0;
        bool busy = false;
    } slot[SLOTS];
    uint64_t submitted = 0, collected = 0;
    PinnedArena arena;
    ~MafChunkStream() {
        for (hipStream_t s : streams)
            if (s) {
                (void)hipStreamSynchronize(s); // (nothing of ours is queued when the buffers go back to the cache)
                (void)hipStreamDestroy(s);
            }
        pinnedArenaGive(arena);
    }
};

static bool mafStreamWanted() {
    const char *e = getenv("HGX_MAF_STREAM");
    return !(e && e[0] == '0');
}

MafChunkStream *mafChunkStreamOpen(hgx_alignment *h, int ref, const ColumnOptions &opt, const std::vector<int32_t> &rankBase, int64_t maxChunk,
                                   int64_t exportColumns, ColumnStats *stats, int64_t uniqueFirst) {
    if (!h->dev || !mafStreamWanted() || maxChunk <= 0 || maxChunk >= (int64_t)LB_COUNT_MAX || rankBase.size() != h->img.genomes.size())
        return nullptr;
    HIP_OK(hipSetDevice(h->dev->device));
    ensureDeviceDna(h->img, *h->dev);
    std::shared_ptr<MafTracks> T = mafTracksFor(h, ref, opt, exportColumns);
    const bool unique = uniqueFirst >= 0;
    if (!T || (unique ? T->stateUnique : T->state).load() == MafTracks::REFUSED)
        return nullptr;
    if (unique && T->constRows == 0)
        return nullptr; // (--unique tells a column's reference bases from its rows: the reference has to be reported)
    std::unique_ptr<MafChunkStream> M(new MafChunkStream);
    M->h = h;
    M->T = T;
    M->ref = ref;
    M->opt = opt;
    M->rankBase = rankBase;
    M->stats = stats;
    M->uniqueFirst = uniqueFirst;
    const char *env = getenv("HGX_MAF_SWEEP");
    M->forced = env && env[0] == '1';
    const uint32_t n = (uint32_t)maxChunk;
    M->maxChunk = n;
    // room: eight rows a column of the chunk for the marked columns' rows on the device; on the host two head rows a column and a
    // head every second column — config 3 needs 0.5, 0.29 and 0.04 of a column; a chunk that needs more is the walk's (Collect: false)
    M->rowsRoom = std::max<uint64_t>(8ull * n, 1u << 20);
    M->outRoom = std::max<uint64_t>(2ull * n, 1u << 16);
    M->headRoom = std::max<uint32_t>(n / 2 + 1, 1u << 12);
    if (const char *e = getenv("HGX_MAF_STREAM_ROOM")) { // (the tests: chunks that do not fit)
        M->outRoom = std::max<uint64_t>(1, (uint64_t)atoll(e));
        M->headRoom = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(M->outRoom, M->headRoom));
    }
    M->tiles1 = (size_t)(n + MAF_MARK_TILE - 1) / MAF_MARK_TILE + 1;
    M->tiles2 = (size_t)(n + 255) / 256 + 1; // (a scan over the marked columns — two of them with --unique —, one over the stretches: as many tiles each at most... see segRoom)
    M->segRoom = unique ? (uint32_t)std::min<uint64_t>(LB_COUNT_MAX - 1, 2ull * n + 1024) : 0; // (a run falls into a few stretches; more: the other launches)
    if (unique)
        M->tiles2 = (size_t)(M->segRoom + 255) / 256 + 1;
    M->P = makeParams(h, ref, 0, 1, 1, opt, nullptr, M->masks);
    M->dRankBase.resize(rankBase.size() * 4);
    HIP_OK(hipMemcpy(M->dRankBase.p, rankBase.data(), rankBase.size() * 4, hipMemcpyHostToDevice));
    for (hipStream_t &st : M->streams)
        HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t bHead = pad((size_t)n + 8), bOff = pad(((size_t)M->headRoom + 1) * 4), bOut = pad((size_t)M->outRoom * sizeof(MafChunkRow)), bCtl = 256,
                 perSlot = bHead + 2 * bOff + bOut + bCtl;
    M->arena = pinnedArenaTake(perSlot * MafChunkStream::SLOTS);
    if (!M->arena.p)
        return nullptr; // (no page-locked memory: the launches that copy)
    char *at = M->arena.p;
    for (MafChunkStream::Slot &S : M->slot) {
        S.candCol.resize((size_t)n * 4);
        S.candRow.resize(((size_t)n + 1) * 4);
        S.rows.resize((size_t)M->rowsRoom * sizeof(ColumnRow));
        S.head.resize((size_t)n + 8);
        S.ctl.resize(64 + (M->tiles1 + 3 * M->tiles2) * 8);
        if (unique)
            S.seg.resize((size_t)M->segRoom * sizeof(UniqueSeg));
        S.headOff.resize(((size_t)M->headRoom + 1) * 4);
        S.headCol.resize(((size_t)M->headRoom + 1) * 4);
        S.out.resize((size_t)M->outRoom * sizeof(MafChunkRow));
        S.hHead.p = at;
        S.hHeadOff.p = at + bHead;
        S.hHeadCol.p = at + bHead + bOff;
        S.hOut.p = at + bHead + 2 * bOff;
        S.hCtl.p = at + bHead + 2 * bOff + bOut;
        at += perSlot;
    }
    return M.release();
}

void mafChunkStreamClose(MafChunkStream *M) {
    delete M;
}

size_t mafChunkStreamInFlight(const MafChunkStream *M) {
    return (size_t)(M->submitted - M->collected);
}

void mafChunkStreamSubmit(MafChunkStream *M, int64_t first, int64_t count) {
    if (count <= 0 || count > (int64_t)M->maxChunk || M->submitted - M->collected >= (uint64_t)MafChunkStream::SLOTS)
        throw std::runtime_error("mafChunkStreamSubmit: no slot for the chunk");
    const GenomeTables &R = M->h->img.genomes[(size_t)M->ref];
    if (first < 0 || first + count > R.totalLength)
        throw std::runtime_error("column range out of bounds for genome " + R.name);
    HIP_OK(hipSetDevice(M->h->dev->device));
    MafChunkStream::Slot &S = M->slot[M->submitted % MafChunkStream::SLOTS];
    const uint32_t n = (uint32_t)count;
    S.first = first;
    S.count = count;
    S.busy = true;
    hipStream_t s = M->streams[M->submitted % MafChunkStream::SLOTS];
    MafChunkCtl *ctl = (MafChunkCtl *)S.ctl.p;
    unsigned long long *tiles1 = (unsigned long long *)((char *)S.ctl.p + 64), *tiles2 = tiles1 + M->tiles1;
    const MafTracks &T = *M->T;
    HIP_OK(hipEventRecord(S.e0.e, s));
    HIP_OK(hipMemsetAsync(S.ctl.p, 0, 64 + (M->tiles1 + 3 * M->tiles2) * 8, s));
    const uint32_t nt1 = (n + MAF_MARK_TILE - 1) / MAF_MARK_TILE;
    hipLaunchKernelGGL(k_maf_mark_list, dim3(nt1), dim3(256), 0, s, T.Fref, T.Aref, T.constRows, first, n, ctl, tiles1, (uint32_t *)S.candCol.p,
                       (uint32_t *)S.candRow.p, (uint8_t *)S.head.p);
    MafRowParams R2;
    R2.P = M->P;
    R2.P.first = first;
    R2.P.count = count;
    R2.S = (const int32_t *const *)T.sPtrs.p;
    R2.refLocate = (const int32_t *)T.refLocate.p;
    R2.refLocateShift = T.refLocateShift;
    R2.candCol = (const uint32_t *)S.candCol.p;
    R2.candRow = (const uint32_t *)S.candRow.p;
    R2.nCand = 0;
    const int rowGrid = (int)std::max<int64_t>(1, std::min<int64_t>(COL_GRID, ((int64_t)n * (1 << MAF_LPC_LOG) / 4 + 255) / 256));
    if (M->h->dev->wide)
        hipLaunchKernelGGL((k_maf_rows_ctl<int64_t>), dim3(rowGrid), dim3(256), 0, s, R2, ctl, (unsigned long long)M->rowsRoom, (ColumnRow *)S.rows.p);
    else
        hipLaunchKernelGGL((k_maf_rows_ctl<int32_t>), dim3(rowGrid), dim3(256), 0, s, R2, ctl, (unsigned long long)M->rowsRoom, (ColumnRow *)S.rows.p);
    const uint32_t nt2 = std::min<uint32_t>((n + 255) / 256, 1024u);
    if (M->uniqueFirst >= 0) {
        if (M->uniqueFirst > first)
            throw std::runtime_error("columnsHeadRowsHost: the range of --unique begins behind its columns");
        UniqueParams U;
        U.candCol = (const uint32_t *)S.candCol.p;
        U.candRow = (const uint32_t *)S.candRow.p;
        U.rows = (const ColumnRow *)S.rows.p;
        U.nCand = 0;
        U.n = n;
        U.first = first;
        U.f = M->uniqueFirst;
        U.ref = M->ref;
        U.maxRefRows = UNIQUE_MAX_REF_ROWS;
        if (const char *e = getenv("HGX_MAF_UNIQUE_MAX_REF"))
            U.maxRefRows = std::max(0, std::min(UNIQUE_MAX_REF_ROWS, atoi(e)));
        U.error = &ctl->error;
        // (the batches that are held against the column walk — the first of a set of tracks, every eighth behind it — keep the walk's
        // form: every column of a stretch that is walked for its keys)
        const bool checked = M->T->stateUnique.load() == MafTracks::CHECKED && M->submitted % 8 != 7;
        S.collapsed = checked && !(getenv("HGX_MAF_UNIQUE_COLLAPSE") && getenv("HGX_MAF_UNIQUE_COLLAPSE")[0] == '0');
        U.collapseKeysOnly = S.collapsed ? 1 : 0;
        unsigned long long *tiles3 = tiles2 + M->tiles2, *tiles4 = tiles3 + M->tiles2;
        hipLaunchKernelGGL(k_unique_stretch_list, dim3(nt2), dim3(256), 0, s, U, ctl, tiles3, (UniqueSeg *)S.seg.p, M->segRoom);
        const uint32_t nt3 = std::min<uint32_t>((M->segRoom + 255) / 256, 1024u);
        hipLaunchKernelGGL(k_unique_out, dim3(nt3), dim3(256), 0, s, U, ctl, tiles4, (const UniqueSeg *)S.seg.p, M->h->dev->desc, (const int32_t *)M->dRankBase.p,
                           M->headRoom, (unsigned long long)M->outRoom, (uint8_t *)S.head.p, (uint32_t *)S.headOff.p, (uint32_t *)S.headCol.p,
                           (MafHeadRow *)S.out.p);
    } else {
        hipLaunchKernelGGL(k_maf_heads_out, dim3(nt2), dim3(256), 0, s, (const uint32_t *)S.candCol.p, (const uint32_t *)S.candRow.p, (const ColumnRow *)S.rows.p,
                           ctl, tiles2, M->h->dev->desc, (const int32_t *)M->dRankBase.p, M->headRoom, (unsigned long long)M->outRoom, (uint8_t *)S.head.p,
                           (uint32_t *)S.headOff.p, (uint32_t *)S.headCol.p, (MafHeadRow *)S.out.p);
    }
    hipLaunchKernelGGL(k_maf_ship, dim3(512), dim3(256), 0, s, ctl, (const uint32_t *)S.headOff.p, (const uint32_t *)S.headCol.p, (const MafHeadRow *)S.out.p,
                       (uint32_t *)S.hHeadOff.p, (uint32_t *)S.hHeadCol.p, (MafHeadRow *)S.hOut.p);
    HIP_OK(hipMemcpyAsync(S.hHead.p, S.head.p, n, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(S.hCtl.p, S.ctl.p, 64, hipMemcpyDeviceToHost, s));
    HIP_OK(hipEventRecord(S.e1.e, s));
    ++M->submitted;
}

// the chunk's heads by the column walk, described and sorted as the stream's are: what a chunk of the stream is held against
static void mafWalkChunk(MafChunkStream *M, int64_t first, int64_t count, std::vector<uint8_t> &head, std::vector<uint32_t> &off, std::vector<MafChunkRow> &rows) {
    HeadRows raw;
    {
        HeadPathOnly only; // (the null stream's path: one at a time)
        columnsHeadRowsWalk(M->h, M->ref, first, count, M->opt, head, off, raw, nullptr, M->uniqueFirst);
    }
    rows.resize(raw.size());
    for (size_t hI = 0; hI + 1 < off.size(); ++hI) {
        const uint32_t a = off[hI], b = off[hI + 1];
        for (uint32_t i = a; i < b; ++i)
            describeHostRow(M->h->img, M->rankBase, raw[i], rows[i], i - a);
        std::stable_sort(rows.begin() + a, rows.begin() + b, [](const MafChunkRow &x, const MafChunkRow &y) { return x.rank < y.rank; });
    }
}

bool mafChunkStreamCollect(MafChunkStream *M, MafChunkOut &out) {
    if (M->collected >= M->submitted)
        throw std::runtime_error("mafChunkStreamCollect: nothing was submitted");
    HIP_OK(hipSetDevice(M->h->dev->device));
    MafChunkStream::Slot &S = M->slot[M->collected % MafChunkStream::SLOTS];
    ++M->collected;
    HIP_OK(hipEventSynchronize(S.e1.e));
    S.busy = false;
    MafChunkCtl ctl;
    memcpy(&ctl, S.hCtl.p, sizeof ctl);
    MafTracks &T = *M->T;
    const bool unique = M->uniqueFirst >= 0;
    std::atomic<int> &state = unique ? T.stateUnique : T.state;
    if (ctl.error == 2) {
        if (M->forced)
            throw MafSizesDoNotAddUp();
        state.store(MafTracks::REFUSED);
        return false;
    }
    if (ctl.error == 1)
        throw std::runtime_error("column walk exceeded the frame stack (more than 64 pending branches for one base)");
    if (ctl.error)
        return false; // (no room, or 2^32 rows: this chunk and what follows by the launches that size their buffers as they go)
    const uint32_t n = (uint32_t)S.count, nHeads = ctl.nHeads;
    const uint64_t total = ctl.totalHeadRows;
    out.n = S.count;
    out.head.assign((const uint8_t *)S.hHead.p, (const uint8_t *)S.hHead.p + n);
    out.headOff.assign((const uint32_t *)S.hHeadOff.p, (const uint32_t *)S.hHeadOff.p + nHeads);
    out.headOff.push_back((uint32_t)total);
    out.headCol.assign((const uint32_t *)S.hHeadCol.p, (const uint32_t *)S.hHeadCol.p + nHeads);
    out.numRows = (size_t)total;
    out.rows = static_cast<MafChunkRow *>(hostBlockTake(std::max<size_t>((size_t)total, 1) * sizeof(MafChunkRow)));
    memcpy(out.rows, S.hOut.p, (size_t)total * sizeof(MafChunkRow));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, S.e0.e, S.e1.e));
    T.deviceUs.fetch_add((uint64_t)(ms * 1e3));
    T.servedColumns.fetch_add((uint64_t)S.count);
    T.servedHeads.fetch_add(nHeads);
    T.servedMarked.fetch_add(ctl.nCand);
    if (M->stats) {
        M->stats->rows_ms += ms;
        M->stats->rows += ctl.totalRows;
        M->stats->columns += (uint64_t)S.count;
    }
    // The first chunk taken from a set of tracks is held against the column walk whole; of every eighth chunk behind it a stretch of
    // 32 k columns somewhere inside it (a structure the first chunk does not hold — an insertion, a ring, a reversed tail — would be
    // wrong without a word otherwise: ADVICE r05).  Tracks that differ are not used again; this chunk and the rest are the walk's.
    const uint64_t index = M->collected - 1;
    const bool whole = state.load() == MafTracks::UNCHECKED;
    if (whole || (index % 8 == 7 && n > 65536 && !S.collapsed)) {
        int64_t a = 0, len = n;
        if (!whole) {
            len = 32768;
            a = (int64_t)((index * 2654435761ull) % (uint64_t)(n - len));
        }
        std::vector<uint8_t> head2;
        std::vector<uint32_t> off2(1, 0);
        std::vector<MafChunkRow> rows2;
        mafWalkChunk(M, S.first + a, len, head2, off2, rows2);
        bool good = true;
        size_t hI = 0; // heads of the chunk in front of column a
        for (int64_t c = 0; c < a; ++c)
            hI += out.head[(size_t)c] & 1;
        size_t h2 = 0;
        const MafChunkRow *mine = static_cast<const MafChunkRow *>(out.rows);
        for (int64_t c = 0; c < len && good; ++c) {
            const bool isHead = (out.head[(size_t)(a + c)] & 1) != 0;
            if (c == 0 && !whole && head2[0] == 1) { // (the walk's stretch begins with a head whatever lies in front of it)
                if (isHead) {
                    const uint32_t x = out.headOff[hI], y = out.headOff[hI + 1], x2 = off2[0], y2 = off2[1];
                    good = y - x == y2 - x2 && memcmp(mine + x, rows2.data() + x2, (size_t)(y - x) * sizeof(MafChunkRow)) == 0;
                    ++hI;
                }
                ++h2;
                continue;
            }
            if (out.head[(size_t)(a + c)] != head2[(size_t)c]) {
                good = false;
                break;
            }
            if (isHead) {
                const uint32_t x = out.headOff[hI], y = out.headOff[hI + 1], x2 = off2[h2], y2 = off2[h2 + 1];
                good = y - x == y2 - x2 && memcmp(mine + x, rows2.data() + x2, (size_t)(y - x) * sizeof(MafChunkRow)) == 0 &&
                       out.headCol[hI] == (uint32_t)(a + c);
                ++hI;
                ++h2;
            }
        }
        if (!good && getenv("HGX_MAF_STREAM_DEBUG")) {
            size_t nh = 0, nh2 = 0;
            for (uint8_t x : out.head)
                nh += x & 1;
            for (uint8_t x : head2)
                nh2 += x & 1;
            fprintf(stderr, "[hgx] stream check: chunk of %u columns at %lld, stretch %lld + %lld: heads %zu (ctl %u, marked %u, rows %llu, head rows %llu) / walk's %zu, rows %zu / %zu\n", n,
                    (long long)S.first, (long long)a, (long long)len, nh, nHeads, ctl.nCand, (unsigned long long)ctl.totalRows, (unsigned long long)total, nh2,
                    (size_t)total, rows2.size());
            size_t i2 = 0, i1 = 0;
            for (int64_t c = 0; c < len; ++c) {
                const bool h1 = out.head[(size_t)(a + c)] & 1, h2b = head2[(size_t)c] & 1;
                if (h1 != h2b) {
                    fprintf(stderr, "[hgx]   column %lld: head %d / %d\n", (long long)c, (int)h1, (int)h2b);
                    break;
                }
                if (h1) {
                    const uint32_t x = out.headOff[i1], y = out.headOff[i1 + 1], x2 = off2[i2], y2 = off2[i2 + 1];
                    if (y - x != y2 - x2 || memcmp(mine + x, rows2.data() + x2, (size_t)(y - x) * sizeof(MafChunkRow)) != 0 || out.headCol[i1] != (uint32_t)(a + c)) {
                        fprintf(stderr, "[hgx]   head %zu at column %lld (headCol %u): rows %u / %u, offsets %u / %u\n", i1, (long long)c, out.headCol[i1], y - x, y2 - x2, x, x2);
                        for (uint32_t q = 0; q < std::max(y - x, y2 - x2) && q < 12; ++q)
                            fprintf(stderr, "[hgx]     row %u: key %lld rank %d ord %u / key %lld rank %d ord %u\n", q, q < y - x ? (long long)mine[x + q].key : -1ll,
                                    q < y - x ? mine[x + q].rank : -1, q < y - x ? mine[x + q].ord : 0u, q < y2 - x2 ? (long long)rows2[x2 + q].key : -1ll,
                                    q < y2 - x2 ? rows2[x2 + q].rank : -1, q < y2 - x2 ? rows2[x2 + q].ord : 0u);
                        break;
                    }
                    ++i1;
                    ++i2;
                }
            }
        }
        if (!good) {
            hostBlockGive(out.rows);
            out.rows = nullptr;
            if (M->forced)
                throw std::runtime_error("hal2maf: the heads taken from the per-base tracks differ from the column walk's");
            state.store(MafTracks::REFUSED);
            fprintf(stderr, "[hgx] hal2maf: the heads taken from the per-base tracks differ from the column walk's; the walk is used\n");
            return false;
        }
        if (whole)
            state.store(MafTracks::CHECKED);
    }
    (unique ? T.chunksUnique : T.chunks).fetch_add(1);
    return true;
}

// what the column engine keeps between calls and nobody is using: the stream's page-locked arenas, the page-locked host blocks
// (hgx_release_cached; a handle's per-base tracks go with hgx_maf_tracks_info(drop) or with the handle)
void columnsReleaseCached() {
    {
        PinnedArenas &A = pinnedArenas();
        std::vector<PinnedArena> idle;
        {
            std::lock_guard<std::mutex> lock(A.mu);
            idle.swap(A.idle);
        }
        for (PinnedArena &a : idle)
            (void)hipHostFree(a.p);
    }
    HostBlockPool &pool = hostBlockPool();
    std::vector<HostBlockHeader *> idle;
    {
        std::lock_guard<std::mutex> lock(pool.mu);
        idle.swap(pool.idle);
        pool.idleBytes = 0;
    }
    for (HostBlockHeader *hd : idle) {
        hd->magic = 0;
        if (hd->pinned)
            (void)hipHostFree(hd);
        else
            free(hd);
    }
}

// ---- hal2maf's text on the device (hgx_maf_render_kernels.hpp) ----
namespace {
// streams of the rendering calls: a call takes one and gives it back (made without the null stream's implicit waits: the device
// stage's launches on the null stream and a batch's rendering have nothing to wait for in each other)
struct RenderStreams {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;
};
RenderStreams &renderStreams() {
    static RenderStreams *r = new RenderStreams; // (never destroyed: the runtime may be gone by then)
    return *r;
}
struct RenderStream {
    int device;
    hipStream_t s = nullptr;
    explicit RenderStream(int dev) : device(dev) {
        RenderStreams &R = renderStreams();
        {
            std::lock_guard<std::mutex> lock(R.mu);
            for (size_t i = 0; i < R.idle.size(); ++i)
                if (R.idle[i].first == dev) {
                    s = R.idle[i].second;
                    R.idle.erase(R.idle.begin() + (std::ptrdiff_t)i);
                    return;
                }
        }
        HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    ~RenderStream() {
        RenderStreams &R = renderStreams();
        std::lock_guard<std::mutex> lock(R.mu);
        R.idle.emplace_back(device, s);
    }
};
} // namespace

bool mafRenderDevice(hgx_alignment *h, const MafRenderInput &in, char *&text, size_t &bytes) {
    text = nullptr;
    bytes = 0;
    if (!h->dev || in.numBlocks == 0 || in.numBlocks >= (1ull << 31) || in.slots >= (1ull << 31) || in.numRows >= (1ull << 32) ||
        in.numRowEnt >= (1ull << 32) || in.numEvents >= (1ull << 32))
        return false;
    HIP_OK(hipSetDevice(h->dev->device));
    if (h->dev->dna.empty() && !h->img.genomes.empty()) {
        bool any = false;
        for (const GenomeTables &G : h->img.genomes)
            any = any || !G.dna.empty();
        if (any)
            return false; // (the bases are not on the device: the export's device stage puts them there; a caller without one renders itself)
    }
    RenderStream stream(h->dev->device);
    hipStream_t s = stream.s;
    const uint32_t nb = (uint32_t)in.numBlocks;
    auto up = [&](Buf &b, const void *src, size_t n) {
        b.resize(std::max<size_t>(n, 16));
        if (n)
            HIP_OK(hipMemcpyAsync(b.p, src, n, hipMemcpyHostToDevice, s));
    };
    Buf dBlocks, dEntRank, dEvents, dRowEnt, dRows, dRanks, dChars;
    up(dBlocks, in.blocks, in.numBlocks * sizeof(MafRenderBlock));
    up(dEntRank, in.entRank, in.numEntRank * 4);
    up(dEvents, in.events, in.numEvents * sizeof(MafRenderEvent));
    up(dRowEnt, in.rowEnt, in.numRowEnt * 4);
    up(dRows, in.rows, in.numRows * sizeof(MafRenderRow));
    up(dRanks, in.ranks, in.numRanks * sizeof(MafRenderRank));
    up(dChars, in.chars, in.numChars);
    Buf dLen((size_t)nb * 4), dOff(((size_t)nb + 1) * 4), dSums(((size_t)nb / SCAN_BLOCK + 2) * 4), dRowOff(std::max<size_t>(in.slots, 1) * 4),
        dSlotBlock(std::max<size_t>(in.slots, 1) * 4), dCtl(16);
    HIP_OK(hipMemsetAsync(dCtl.p, 0, 16, s));
    MafRenderParams P;
    P.blocks = (const MafRenderBlock *)dBlocks.p;
    P.entRank = (const int32_t *)dEntRank.p;
    P.events = (const MafRenderEvent *)dEvents.p;
    P.rowEnt = (const uint32_t *)dRowEnt.p;
    P.rows = (const MafRenderRow *)dRows.p;
    P.ranks = (const MafRenderRank *)dRanks.p;
    P.chars = (const char *)dChars.p;
    P.desc = h->dev->desc;
    P.numBlocks = nb;
    P.slots = (uint32_t)in.slots;
    P.keepEmptyRefBlocks = in.keepEmptyRefBlocks ? 1 : 0;
    P.blockLen = (uint32_t *)dLen.p;
    P.blockOff = (const uint32_t *)dOff.p;
    P.rowOff = (uint32_t *)dRowOff.p;
    P.slotBlock = (uint32_t *)dSlotBlock.p;
    P.total = (unsigned long long *)dCtl.p;
    P.error = (unsigned int *)((char *)dCtl.p + 8);
    P.text = nullptr;
    const int gridB = (int)std::max<int64_t>(1, std::min<int64_t>(COL_GRID, ((int64_t)nb + 255) / 256));
    const int gridS0 = (int)std::max<int64_t>(1, std::min<int64_t>(2 * COL_GRID, ((int64_t)in.slots + 255) / 256));
    hipLaunchKernelGGL(k_maf_render_blocks, dim3(gridB), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_maf_render_rowlen, dim3(gridS0), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_maf_render_sizes, dim3(gridB), dim3(256), 0, s, P);
    {
        const uint32_t tiles = (nb + SCAN_BLOCK - 1) / SCAN_BLOCK;
        hipLaunchKernelGGL(k_scan_block_sums, dim3(tiles), dim3(256), 0, s, (const uint32_t *)dLen.p, nb, (uint32_t *)dSums.p);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, (uint32_t *)dSums.p, tiles, (uint32_t *)dOff.p + nb);
        hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(256), 0, s, (const uint32_t *)dLen.p, nb, (const uint32_t *)dSums.p, (uint32_t *)dOff.p);
    }
    struct {
        unsigned long long total;
        unsigned int error, pad;
    } ctl{0, 0, 0};
    HIP_OK(hipMemcpyAsync(&ctl, dCtl.p, 16, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    if (ctl.error || ctl.total >= (1ull << 32))
        return false; // (offsets of 32 bits: a batch of four gigabytes of text is the rendering threads')
    bytes = (size_t)ctl.total;
    if (bytes == 0) {
        text = static_cast<char *>(hostBlockTake(1));
        return true;
    }
    Buf dText(bytes);
    P.text = (char *)dText.p;
    const int gridS = (int)std::max<int64_t>(1, std::min<int64_t>(2 * COL_GRID, ((int64_t)in.slots + 255) / 256));
    hipLaunchKernelGGL(k_maf_render_rows, dim3(gridS), dim3(256), 0, s, P);
    char *host = static_cast<char *>(hostBlockTake(bytes));
    hipError_t e = hipMemcpyAsync(host, dText.p, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        hostBlockGive(host);
        throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " while a batch's MAF text was rendered on the device");
    }
    text = host;
    return true;
}

} // namespace hgx
