// halAlignmentDepth's values as wig lines ("%d\n" each, alignmentDepth/halAlignmentDepth.cpp:246, 271, 305): the device scans a
// quarter of a billion columns in a hundredth of a second, so the text must not be made by one thread calling snprintf per
// column (fifty nanoseconds each).  The lines' sizes are counted and the lines written by as many threads as the host lends,
// straight into the output where the stream gives room for them at once (BulkSink, hgx_columns_host.hpp).
#pragma once
#include <algorithm>
#include <charconv>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <future>
#include <mutex>
#include <ostream>
#include <string>
#include <thread>
#include <vector>
#include "hgx_host_threads.hpp"

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
    unsigned nt = hostThreads();
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
    char *dst = nullptr;
    auto write = [&](unsigned t) {
        char *o = dst + bytes[t];
        for (int64_t i = part(t); i < part(t + 1); ++i)
            o = wigLine(o, vals[i]);
    };
    std::string own;
    auto place = [&]() { // every part's place in the text, and the text's
        for (unsigned t = 0; t < nt; ++t)
            bytes[t + 1] += bytes[t];
        dst = room(bytes[nt]);
        if (!dst) {
            own.resize(bytes[nt]);
            dst = &own[0];
        }
    };
    if (nt == 1) {
        measure(0);
        place();
        write(0);
    } else {
        // the threads are made once: each counts its part, waits until all have and the places are known, writes its part
        std::mutex mu;
        std::condition_variable cv;
        unsigned counted = 0;
        bool placed = false, failed = false;
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t)
            th.emplace_back([&, t]() {
                measure(t);
                std::unique_lock<std::mutex> lock(mu);
                ++counted;
                cv.notify_all();
                cv.wait(lock, [&]() { return placed || failed; });
                lock.unlock();
                if (!failed)
                    write(t);
            });
        measure(0);
        {
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&]() { return counted == nt - 1; });
            try {
                place();
                placed = true;
            } catch (...) { // (no memory for the text: the threads must not be left waiting)
                failed = true;
                cv.notify_all();
                lock.unlock();
                for (std::thread &x : th)
                    x.join();
                throw;
            }
            cv.notify_all();
        }
        write(0);
        for (std::thread &x : th)
            x.join();
    }
    if (!own.empty())
        os.write(own.data(), (std::streamsize)own.size());
}

// count values in chunks through two buffers in turn: copy(buffer, first, n) fills a buffer with values [first, first + n),
// sink(buffer, first, n) takes them — in order, one at a time, the copy of a chunk going on while the sink has the chunk before
// (the device's copies into page-locked blocks beside the threads that make the lines: hgx_columns.hip, columnsDepthChunksHost).
template <class Copy, class Sink> void handOffChunks(int32_t *const buffer[2], int64_t count, int64_t chunk, Copy copy, Sink sink) {
    std::future<void> pending; // (waited for by its destructor too: nothing of the caller's is let go while a sink runs)
    int turn = 0;
    for (int64_t lo = 0; lo < count; lo += chunk, turn ^= 1) {
        const int64_t n = std::min(chunk, count - lo);
        int32_t *p = buffer[turn]; // (the sink that read this buffer last was waited for a round ago)
        copy(p, lo, n);
        if (pending.valid())
            pending.get(); // (in order: a sink appends to the text where the one before stopped)
        pending = std::async(std::launch::async, [&sink, p, lo, n]() { sink(p, lo, n); });
    }
    if (pending.valid())
        pending.get();
}

} // namespace hgx
