// The profiling build's stand-in for the device in halLiftover's host code (make hostprof-lib; not part of libhgx.so):
// HGX_LIFT_REPLAY names a file of hgx_record rows, the index of the interval counted over the whole conversion, as
// `hal_oracle liftover --records` writes them; a batch of n intervals takes the rows of the next n indices.  So the host side of
// both text paths (hgx_liftover_host.cpp: BED12, PSL, mixed column counts; hgx_liftover_text.cpp: the parallel one) runs against
// the oracle's text on a machine without a GPU.
#pragma once
#ifdef HGX_HOST_PROFILE
#include "../../include/hgx.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace hgx {

struct LiftReplay {
    FILE *f = nullptr;
    hgx_record next{};
    bool have = false;
    int64_t base = 0; // intervals of the batches before
    LiftReplay() {
        if (const char *p = getenv("HGX_LIFT_REPLAY"))
            f = fopen(p, "rb");
    }
    void batch(size_t n, std::vector<hgx_record> &recs) {
        recs.clear();
        for (;;) {
            if (!have)
                have = fread(&next, sizeof next, 1, f) == 1;
            if (!have || next.query >= base + (int64_t)n)
                break;
            if (next.query < base)
                throw std::runtime_error("HGX_LIFT_REPLAY: the file does not continue with this batch");
            recs.push_back(next);
            recs.back().query -= base;
            have = false;
        }
        base += (int64_t)n;
    }
    // the rows of the intervals [first, first + n) of the conversion, their indices counted from `first` (the parallel text path:
    // its groups of chunks are lifted side by side, by several handles; one conversion a process)
    std::vector<hgx_record> all;
    std::once_flag loaded;
    void range(int64_t first, size_t n, std::vector<hgx_record> &recs) {
        std::call_once(loaded, [this]() {
            hgx_record r;
            while (fread(&r, sizeof r, 1, f) == 1)
                all.push_back(r);
        });
        auto lo = std::lower_bound(all.begin(), all.end(), first, [](const hgx_record &a, int64_t q) { return a.query < q; });
        auto hi = std::lower_bound(lo, all.end(), first + (int64_t)n, [](const hgx_record &a, int64_t q) { return a.query < q; });
        recs.assign(lo, hi);
        for (hgx_record &r : recs)
            r.query -= first;
    }
};
inline LiftReplay &liftReplay() {
    static LiftReplay r;
    return r;
}

} // namespace hgx
#endif
