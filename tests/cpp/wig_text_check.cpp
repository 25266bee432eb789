// halAlignmentDepth's wig lines (hal_amd/csrc/hgx_wig_text.hpp: sizes counted and lines written by many threads) against
// snprintf("%d\n") line by line: every digit count, negatives, INT32_MIN, 3 M random values, 1 / 3 / 64 threads, both ways out
// (into room the stream gives, through ostream::write).  Built and run by tests/test_capi_host.py.
#include "hgx_wig_text.hpp"
#include <chrono>
#include <cstdio>
#include <random>
#include <sstream>
#include <stdexcept>
int main() {
    std::mt19937_64 rng(7);
    std::vector<int32_t> v;
    const int32_t edge[] = {0, 1, 9, 10, 11, 99, 100, 101, 999, 1000, 9999, 10000, 99999, 100000, 999999, 1000000, 9999999, 10000000, 99999999, 100000000, 999999999, 1000000000, 2147483647, -1, -9, -10, -99, -100, -2147483647, (int32_t)0x80000000};
    for (int32_t e : edge) v.push_back(e);
    for (int i = 0; i < 3000000; ++i) { int k = rng() % 4; v.push_back(k == 0 ? (int32_t)(rng() % 10) : k == 1 ? (int32_t)(rng() % 200) : k == 2 ? (int32_t)(rng() % 100000) : (int32_t)rng()); }
    std::string want; char tmp[16];
    for (int32_t x : v) { int n = snprintf(tmp, sizeof tmp, "%d\n", x); want.append(tmp, n); }
    for (unsigned threads : {1u, 3u, 64u}) {
        std::ostringstream os; hgx::wigLines(os, v.data(), (int64_t)v.size(), [](size_t) { return (char *)nullptr; }, threads);
        if (os.str() != want) { printf("DIFFERENT (stream, %u threads)\n", threads); return 1; }
        std::string buf; std::ostringstream os2; hgx::wigLines(os2, v.data(), (int64_t)v.size(), [&](size_t n) { buf.resize(n); return &buf[0]; }, threads);
        if (buf != want || !os2.str().empty()) { printf("DIFFERENT (room, %u threads)\n", threads); return 1; }
    }
    for (int64_t n : {0, 1, 2, 65535, 65536, 65537}) { std::ostringstream os; hgx::wigLines(os, v.data(), n, [](size_t) { return (char *)nullptr; }); std::string w; for (int64_t i = 0; i < n; ++i) { int k = snprintf(tmp, sizeof tmp, "%d\n", v[i]); w.append(tmp, k); } if (os.str() != w) { printf("DIFFERENT n=%ld\n", (long)n); return 1; } }
    // the chunks' hand-off (columnsDepthChunksHost's loop with a memcpy in the device copy's place): every value once, in order,
    // whatever the chunk size; the text through it is the text at once; a sink's exception comes out
    for (int64_t chunk : {1, 7, 4096, 65536, 1000000, 5000000}) {
        const int64_t count = chunk == 1 ? 1000 : chunk == 7 ? 20000 : (int64_t)v.size();
        std::vector<int32_t> a((size_t)std::min(chunk, count)), b((size_t)std::min(chunk, count));
        int32_t *const buffer[2] = {a.data(), b.data()};
        std::ostringstream os;
        int64_t next = 0;
        bool ordered = true;
        hgx::handOffChunks(buffer, count, chunk, [&](int32_t *p, int64_t lo, int64_t n) { memcpy(p, v.data() + lo, (size_t)n * 4); },
                           [&](const int32_t *p, int64_t lo, int64_t n) {
                               ordered = ordered && lo == next;
                               next = lo + n;
                               hgx::wigLines(os, p, n, [](size_t) { return (char *)nullptr; }, 3);
                           });
        std::string w;
        for (int64_t i = 0; i < count; ++i) { int k = snprintf(tmp, sizeof tmp, "%d\n", v[i]); w.append(tmp, k); }
        if (!ordered || next != count || os.str() != w) { printf("DIFFERENT (chunks of %ld)\n", (long)chunk); return 1; }
    }
    {
        std::vector<int32_t> a(10), b(10);
        int32_t *const buffer[2] = {a.data(), b.data()};
        bool caught = false;
        try {
            hgx::handOffChunks(buffer, 100, 10, [&](int32_t *p, int64_t lo, int64_t n) { memcpy(p, v.data() + lo, (size_t)n * 4); },
                               [&](const int32_t *, int64_t lo, int64_t) { if (lo == 30) throw std::runtime_error("sink"); });
        } catch (const std::runtime_error &) { caught = true; }
        if (!caught) { printf("a sink's exception was lost\n"); return 1; }
    }
    printf("same\n");
    return 0;
}
