// The profiling build's stand-in for the device in halLiftover's host code (make hostprof-lib; not part of libhgx.so):
// HGX_LIFT_REPLAY names a file of hgx_record rows, the index of the interval counted over the whole conversion, as
// `hal_oracle liftover --records` writes them; a batch of n intervals takes the rows of the next n indices.  So the host side of
// both text paths (hgx_liftover_host.cpp: BED12, PSL, mixed column counts; hgx_liftover_text.cpp: the parallel one) runs against
// the oracle's text on a machine without a GPU.
#pragma once
#ifdef HGX_HOST_PROFILE
#include "../../include/hgx.h"
#include <cstdio>
#include <cstdlib>
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
};
inline LiftReplay &liftReplay() {
    static LiftReplay r;
    return r;
}

} // namespace hgx
#endif
