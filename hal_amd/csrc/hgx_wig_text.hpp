// halAlignmentDepth's values as wig lines ("%d\n" each, alignmentDepth/halAlignmentDepth.cpp:246, 271, 305): the device scans a
// quarter of a billion columns in a hundredth of a second, so the text must not be made by one thread calling snprintf per
// column (fifty nanoseconds each).  The lines' sizes are counted and the lines written by as many threads as the host lends,
// straight into the output where the stream gives room for them at once (BulkSink, hgx_columns_host.hpp).
#pragma once
#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstring>
#include <ostream>
#include <string>
#include <thread>
#include <vector>

namespace hgx {

inline size_t wigLineLength(int32_t v) { // bytes of "%d\n"
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    size_t n = v < 0 ? 3 : 2;
    while (u >= 10) {
        u /= 10;
        ++n;
    }
    return n;
}

inline char *wigLine(char *o, int32_t v) {
    if ((uint32_t)v < 10) {
        *o++ = (char)('0' + v);
    } else if ((uint32_t)v < 100) {
        *o++ = (char)('0' + v / 10);
        *o++ = (char)('0' + v % 10);
    } else {
        o = std::to_chars(o, o + 11, v).ptr;
    }
    *o++ = '\n';
    return o;
}

// room(bytes): where that many bytes of text go (counted as written), or null: then the lines go through os.write
template <class Room> void wigLines(std::ostream &os, const int32_t *vals, int64_t count, Room room, unsigned maxThreads = 64) {
    if (count <= 0)
        return;
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt ? nt : 1u, maxThreads));
    nt = (unsigned)std::min<int64_t>(nt, (count + 65535) / 65536); // (a thread per 64 k lines at least)
    std::vector<size_t> bytes(nt + 1, 0);
    auto part = [&](unsigned t) { return count * t / nt; };
    auto measure = [&](unsigned t) {
        size_t n = 0;
        for (int64_t i = part(t); i < part(t + 1); ++i)
            n += (uint32_t)vals[i] < 10 ? 2 : wigLineLength(vals[i]);
        bytes[t + 1] = n;
    };
    auto spread = [&](auto &&fn) {
        if (nt == 1) {
            fn(0u);
            return;
        }
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t)
            th.emplace_back(fn, t);
        fn(0u);
        for (std::thread &x : th)
            x.join();
    };
    spread(measure);
    for (unsigned t = 0; t < nt; ++t)
        bytes[t + 1] += bytes[t];
    const size_t total = bytes[nt];
    std::string own;
    char *dst = room(total);
    if (!dst) {
        own.resize(total);
        dst = &own[0];
    }
    auto write = [&](unsigned t) {
        char *o = dst + bytes[t];
        for (int64_t i = part(t); i < part(t + 1); ++i)
            o = wigLine(o, vals[i]);
    };
    spread(write);
    if (!own.empty())
        os.write(own.data(), (std::streamsize)own.size());
}

} // namespace hgx
