// The text path of Liftover::convert (liftover/impl/halLiftover.cpp:23-92) for what halLiftover is mostly fed: BED files whose
// lines all have the same number of columns, up to nine, lifted to BED.  One pass over the input buffer per stage, every stage
// spread over the host's cores:
//   1. the buffer is cut into chunks at line ends; every chunk is tokenised in place (no copies of lines or fields): sequence
//      look-up, coordinates, strand, the numeric fields the output re-prints, and where the fields start that it echoes;
//   2. the intervals go into pinned arrays in genome coordinates, one H2D copy, the kernels, one D2H copy of the records into
//      pinned memory (liftoverBatchStaged);
//   3. every chunk's output lines are rendered from its records into the chunk's own slice of the one output buffer.
// Semantics are those of BedScanner::scan / BedLine::read / Liftover::visitLine / BedLine::write
// (halBedScanner.cpp:40-61, halBedLine.cpp:27-151, halLiftover.cpp:46-92): blank space between lines is skipped and does not
// count as a line, a line's fields are split at tabs (empty fields kept, a trailing tab adds none), integers read like
// operator>> (leading blanks, optional sign, digits, the rest ignored), score / thick range / itemRgb are re-printed from
// their values, every other field is echoed, skipped lines (unknown sequence, end past the sequence) produce nothing, and the
// first malformed line ends the conversion with the reference's message after the lines before it have been written.
// Anything else — BED12 lines, PSL output, files that mix column counts (the reference's line object then inherits fields from
// earlier lines) — returns false and takes the general path of hgx_liftover_host.cpp.
#include "hgx_textmem.hpp"
#include "hgx_liftover_host.hpp"
#include "hgx_lift_replay.hpp"
#include <mutex>
#include <exception>
#include <functional>
#include <condition_variable>
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>
#include <unordered_map>

namespace hgx {

namespace {

struct Line {
    const char *text;  // the line (without its newline)
    uint32_t len;
    uint32_t nameOff, nameLen;   // field 3
    uint32_t extraOff, extraEnd; // the echoed fields after the parsed ones: text[extraOff, extraEnd); extraOff == NO_EXTRA: none
    int32_t seq;                 // source sequence, or -1: skipped
    int64_t start, end, score, thickStart, thickEnd, r, g, b;
    char strand;
    int64_t query;               // index of the interval in the batch, -1: skipped
};

constexpr uint32_t NO_EXTRA = 0xFFFFFFFFu;

struct Chunk {
    const char *begin, *end;
    std::vector<Line> lines;
    size_t numQueries = 0, firstQuery = 0, firstLine = 0;
    int device = 0;               // which handle lifts the chunk's lines (firstQuery counts inside that handle's batch)
    std::string error;            // first malformed line of the chunk (then `lines` ends before it)
    size_t errorLine = 0;         // its number inside the chunk, 1-based
    size_t linesSeen = 0;
    bool general = false;         // a line the fast path does not handle
    int bedType = -1;             // column count of the chunk's lines
    std::vector<std::string> notes; // what the reference writes to stderr, in order
    std::string out;             // its lines (only where they cannot be written into the output text directly)
    size_t outBytes = 0, outAt = 0; // how long its lines are, where they stand in the output text
};

inline bool isBlank(char c) { // std::isspace in the "C" locale
    return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
}

// operator>>(int64) on a field: leading blanks, optional sign, digits; stops at the first other character
bool readInt(const char *b, const char *e, int64_t &v) {
    while (b < e && isBlank(*b))
        ++b;
    bool neg = false;
    if (b < e && (*b == '+' || *b == '-')) {
        neg = *b == '-';
        ++b;
    }
    if (b >= e || *b < '0' || *b > '9')
        return false;
    uint64_t x = 0;
    const uint64_t lim = neg ? (uint64_t)1 << 63 : ((uint64_t)1 << 63) - 1;
    for (; b < e && *b >= '0' && *b <= '9'; ++b) {
        const uint64_t d = (uint64_t)(*b - '0');
        if (x > (lim - d) / 10)
            return false; // out of range
        x = x * 10 + d;
    }
    v = neg ? (int64_t)(0 - x) : (int64_t)x;
    return true;
}

struct Fail {
    std::string what;
};

inline int64_t intField(const char *b, const char *e) {
    int64_t v;
    if (!readInt(b, e, v))
        throw Fail{"Error converting string to int: " + std::string(b, e)};
    return v;
}

inline char *putInt(char *p, int64_t v) {
    return std::to_chars(p, p + 24, v).ptr;
}

void parseChunk(Chunk &C, int forcedType, const std::unordered_map<std::string, int> &seqByName, const GenomeTables &S) {
    const char *p = C.begin;
    std::string key;
    bool haveKey = false; // (key is the sequence name of the line before, found at keySeq)
    std::unordered_map<std::string, int>::const_iterator keySeq = seqByName.end();
    C.lines.reserve((size_t)(C.end - C.begin) / 32 + 16); // (a line of six columns is some forty bytes: the vector seldom has to move)
    while (true) {
        while (p < C.end && isBlank(*p)) // BedScanner::skipWhiteSpaces
            ++p;
        if (p >= C.end)
            break;
        const char *eol = (const char *)memchr(p, '\n', (size_t)(C.end - p));
        if (!eol)
            eol = C.end;
        ++C.linesSeen;
        Line L{};
        L.text = p;
        L.len = (uint32_t)(eol - p);
        try {
            // fields (chopString: empty fields kept, nothing after a trailing separator)
            const char *fb[13], *fe[13];
            int nf = 0;
            const char *q = p;
            bool moreFields = false; // more than twelve
            while (q < eol) {
                const char *t = (const char *)memchr(q, '\t', (size_t)(eol - q));
                if (nf == 12) {
                    moreFields = true;
                    break;
                }
                fb[nf] = q;
                fe[nf] = t ? t : eol;
                ++nf;
                if (!t)
                    break;
                q = t + 1;
            }
            auto whole = [&]() { return std::string(p, eol); };
            if (nf < 3)
                throw Fail{"Expected at least three columns in BED record: " + whole()};
            const int bt = forcedType ? forcedType : nf; // (min(columns, 12): nf stops at twelve)
            if (bt > 9 || bt == 7 || (C.bedType >= 0 && bt != C.bedType)) { // BED12, or fields inherited between lines: general path
                C.general = true;
                return;
            }
            C.bedType = bt;
            if (bt > nf) // (the reference reads past the end of its row here)
                throw Fail{"Expected at least " + std::to_string(bt) + " columns in BED record: " + whole()};
            L.start = intField(fb[1], fe[1]);
            L.end = intField(fb[2], fe[2]);
            if (L.start >= L.end)
                throw Fail{"Error zero or negative length BED range: " + whole()};
            L.strand = '+';
            if (bt > 3) {
                L.nameOff = (uint32_t)(fb[3] - p);
                L.nameLen = (uint32_t)(fe[3] - fb[3]);
            }
            if (bt > 4)
                L.score = intField(fb[4], fe[4]);
            if (bt > 5) {
                L.strand = fb[5] < fe[5] ? *fb[5] : '\0';
                if (L.strand != '.' && L.strand != '+' && L.strand != '-')
                    throw Fail{"Strand character must be + or - or ." + whole()};
            }
            if (bt > 6)
                L.thickStart = intField(fb[6], fe[6]);
            if (bt > 7)
                L.thickEnd = intField(fb[7], fe[7]);
            if (bt > 8) {
                const char *c1 = (const char *)memchr(fb[8], ',', (size_t)(fe[8] - fb[8]));
                const char *c2 = c1 ? (const char *)memchr(c1 + 1, ',', (size_t)(fe[8] - c1 - 1)) : nullptr;
                const char *c3 = c2 ? (const char *)memchr(c2 + 1, ',', (size_t)(fe[8] - c2 - 1)) : nullptr;
                // chopString on ',': empty field, or more than three parts (a third comma followed by anything) is an error
                if (fb[8] == fe[8] || (c3 && c3 + 1 < fe[8]))
                    throw Fail{"Error parsing BED itemRGB: " + whole()};
                L.r = intField(fb[8], c1 ? c1 : fe[8]);
                L.g = L.b = L.r;
                if (c1 && c1 + 1 < fe[8])
                    L.g = intField(c1 + 1, c2 ? c2 : fe[8]);
                if (c2 && c2 + 1 < fe[8])
                    L.b = intField(c2 + 1, c3 ? c3 : fe[8]);
            }
            // the fields after the parsed ones are echoed as they are: from the start of field bt to the end of the last field
            // (chopString: a tab at the very end of the line starts no field)
            L.extraOff = NO_EXTRA;
            if (bt < nf || moreFields) {
                L.extraOff = (uint32_t)(fb[bt] - p);
                L.extraEnd = (uint32_t)((eol[-1] == '\t' ? eol - 1 : eol) - p);
            }
            // Liftover::visitLine (halLiftover.cpp:52-66)
            // (the lines of a file mostly name one sequence after the other: the name is looked up when it changes)
            if (!haveKey || key.size() != (size_t)(fe[0] - fb[0]) || memcmp(key.data(), fb[0], key.size()) != 0) {
                key.assign(fb[0], fe[0]);
                keySeq = seqByName.find(key);
                haveKey = true;
            }
            const auto it = keySeq;
            L.seq = -1;
            if (it == seqByName.end()) {
                C.notes.push_back("?" + key);
            } else if (L.end > S.seqs[(size_t)it->second].length) {
                C.notes.push_back("Skipping interval with endpoint " + std::to_string(L.end) + "because sequence " + key + " has length " +
                                  std::to_string(S.seqs[(size_t)it->second].length));
            } else {
                L.seq = it->second;
                ++C.numQueries;
            }
        } catch (const Fail &f) {
            C.error = f.what;
            C.errorLine = C.linesSeen;
            return;
        }
        C.lines.push_back(L);
        p = eol < C.end ? eol + 1 : C.end;
    }
}

// what a line needs of a record, from the record itself or from its 8-byte form (liftoverBatchStaged: PackedRecords)
struct RecView {
    int64_t tgt_start, tgt_end;
    int32_t tgt_seq;
    char strand;
};
struct RecCursor {
    const hgx_record *r = nullptr, *rEnd = nullptr; // the records as they are, sorted by query ...
    const uint32_t *words = nullptr, *first = nullptr; // ... or packed, with every interval's first record
    size_t at = 0, end = 0;
    void seek(int64_t query) {
        if (words) {
            at = first[query];
            end = first[query + 1];
        }
    }
    bool next(int64_t query, RecView &v) {
        if (words) {
            if (at == end)
                return false;
            const uint32_t a = words[2 * at], b = words[2 * at + 1];
            ++at;
            v.tgt_start = (int64_t)a;
            v.tgt_end = (int64_t)a + (int64_t)(b & ((1u << 22) - 1u));
            v.tgt_seq = (int32_t)((b >> 22) & 127u);
            v.strand = "+-."[(b >> 29) & 3u];
            return true;
        }
        if (r == rEnd || r->query != query)
            return false;
        v.tgt_start = r->tgt_start;
        v.tgt_end = r->tgt_end;
        v.tgt_seq = r->tgt_seq;
        v.strand = r->strand;
        ++r;
        return true;
    }
};

// BedLine::write (halBedLine.cpp:104-151) with what BlockLiftover::liftInterval and Liftover::cleanResults substitute.  One body
// for two sinks: the lines are counted first, then written at their place in the one output text (the chunks' texts used to be
// made apart and copied together: as many bytes moved again as rendered).
struct CountSink {
    size_t n = 0;
    void bytes(const char *, size_t k) { n += k; }
    void ch(char) { ++n; }
    void num(int64_t v) { // what putInt writes
        uint64_t u = v < 0 ? 0 - (uint64_t)v : (uint64_t)v;
        size_t k = v < 0 ? 2 : 1;
        for (uint64_t p = 10; u >= p; p *= 10) {
            ++k;
            if (p > UINT64_MAX / 10)
                break;
        }
        n += k;
    }
};
struct WriteSink {
    char *w;
    void bytes(const char *p, size_t k) {
        memcpy(w, p, k);
        w += k;
    }
    void ch(char c) { *w++ = c; }
    void num(int64_t v) { w = putInt(w, v); }
};
template <class Sink> void emitChunk(const Chunk &C, const hgx_record *recs, size_t nRecs, const PackedRecords &packed, const GenomeTables &T, Sink &o) {
    if (C.numQueries == 0)
        return;
    RecCursor cur;
    if (packed.words) {
        cur.words = packed.words;
        cur.first = packed.first;
    } else {
        // the chunk's records: queries [firstQuery, firstQuery + numQueries), the records are sorted by query
        cur.r = std::lower_bound(recs, recs + nRecs, (int64_t)C.firstQuery, [](const hgx_record &a, int64_t q) { return a.query < q; });
        cur.rEnd = recs + nRecs;
    }
    const int bt = C.bedType;
    RecView rec{};
    for (const Line &L : C.lines) {
        if (L.query < 0)
            continue;
        const bool thick = bt > 6 && (L.thickStart != 0 || L.thickEnd != 0);
        cur.seek(L.query);
        for (const RecView *r = &rec; cur.next(L.query, rec);) {
            const std::string &chrom = T.seqs[(size_t)r->tgt_seq].name;
            o.bytes(chrom.data(), chrom.size());
            o.ch('\t');
            o.num(r->tgt_start);
            o.ch('\t');
            o.num(r->tgt_end);
            if (bt > 3) {
                o.ch('\t');
                o.bytes(L.text + L.nameOff, L.nameLen);
            }
            if (bt > 4) {
                o.ch('\t');
                o.num(L.score);
            }
            if (bt > 5) {
                o.ch('\t');
                o.ch(r->strand);
            }
            if (bt > 6) { // (bt == 7 never gets here)
                o.ch('\t');
                o.num(thick ? r->tgt_start : L.thickStart);
            }
            if (bt > 7) {
                o.ch('\t');
                o.num(thick ? r->tgt_end : L.thickEnd);
            }
            if (bt > 8) {
                o.ch('\t');
                o.num(L.r);
                o.ch(',');
                o.num(L.g);
                o.ch(',');
                o.num(L.b);
            }
            if (L.extraOff != NO_EXTRA) {
                o.ch('\t');
                o.bytes(L.text + L.extraOff, L.extraEnd - L.extraOff);
            }
            o.ch('\n');
        }
    }
}
size_t measureChunk(const Chunk &C, const hgx_record *recs, size_t nRecs, const PackedRecords &packed, const GenomeTables &T) {
    CountSink n;
    emitChunk(C, recs, nRecs, packed, T, n);
    return n.n;
}
void renderChunk(const Chunk &C, const hgx_record *recs, size_t nRecs, const PackedRecords &packed, const GenomeTables &T, char *dst, size_t bytes) {
    WriteSink w{dst};
    emitChunk(C, recs, nRecs, packed, T, w);
    if ((size_t)(w.w - dst) != bytes) // (the two sinks share one body: this cannot be; better loud than a text with a hole)
        throw std::logic_error("hgx_liftover_text: a chunk's lines are not as long as they were counted");
}

// The threads of the text path's phases are kept (a call has three phases of a few milliseconds each: thirty-one threads made and
// joined per phase were a fifth of the call).  One job at a time: a caller that finds the pool at work makes threads of its own.
struct TextPool {
    std::mutex jobMu, mu;
    std::condition_variable wake, done;
    std::vector<std::thread> workers;
    const std::function<void()> *work = nullptr;
    unsigned generation = 0, want = 0, running = 0;
    std::exception_ptr failure; // the first exception of a worker of the job under way
    void worker(unsigned idx) {
        unsigned seen = 0;
        for (;;) {
            const std::function<void()> *w = nullptr;
            {
                std::unique_lock<std::mutex> lock(mu);
                wake.wait(lock, [&] { return generation != seen; });
                seen = generation;
                if (idx < want)
                    w = work;
            }
            if (!w)
                continue;
            std::exception_ptr err;
            try {
                (*w)();
            } catch (...) { // (handed to the caller of run: an exception that leaves a detached thread ends the process)
                err = std::current_exception();
            }
            std::lock_guard<std::mutex> lock(mu);
            if (err && !failure)
                failure = err;
            if (--running == 0)
                done.notify_all();
        }
    }
    // work() on `threads` threads, this one among them; false: the pool is busy (nothing was run)
    bool run(unsigned threads, const std::function<void()> &w) {
        std::unique_lock<std::mutex> job(jobMu, std::try_to_lock);
        if (!job.owns_lock())
            return false;
        if (threads > 1) {
            std::lock_guard<std::mutex> lock(mu);
            while (workers.size() + 1 < threads) {
                const unsigned idx = (unsigned)workers.size();
                workers.emplace_back([this, idx] { worker(idx); });
                workers.back().detach();
            }
            work = &w;
            want = threads - 1;
            running = want;
            ++generation;
        }
        if (threads > 1)
            wake.notify_all();
        // (w and what it captured live on the caller's stack: whatever happens on this thread, the workers are waited for
        // before run is left)
        std::exception_ptr err;
        try {
            w();
        } catch (...) {
            err = std::current_exception();
        }
        if (threads > 1) {
            std::unique_lock<std::mutex> lock(mu);
            done.wait(lock, [&] { return running == 0; });
            if (!err)
                err = failure;
            failure = nullptr;
        }
        if (err)
            std::rethrow_exception(err);
        return true;
    }
};
TextPool &textPool() {
    static TextPool *p = new TextPool; // (never destroyed: its threads wait for work until the process ends)
    return *p;
}

template <typename F> void forEachChunk(std::vector<Chunk> &chunks, unsigned threads, F f) {
    std::atomic<size_t> next{0};
    const std::function<void()> work = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < chunks.size();)
            f(chunks[i]);
    };
    if (textPool().run(threads, work))
        return;
    // (the pool is busy: threads of this call's own, joined whatever happens — an exception of one of them comes out here)
    std::mutex errMu;
    std::exception_ptr err;
    const auto guarded = [&]() {
        try {
            work();
        } catch (...) {
            std::lock_guard<std::mutex> lock(errMu);
            if (!err)
                err = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < threads; ++t)
            pool.emplace_back(guarded);
    } catch (...) { // (a thread could not be made: the ones there are do the work)
    }
    guarded();
    for (std::thread &t : pool)
        t.join();
    if (err)
        std::rethrow_exception(err);
}

} // namespace

bool liftoverTextFast(hgx_alignment *const *als, int nAls, int srcGenome, const char *text, size_t len, int tgtGenome, int bedType,
                      bool traverseDupes, int coalescenceLimit, char **outText, size_t *outLen, std::string &error,
                      std::set<std::string> &missedSet, hgx_liftover_stats &stats, size_t batchLines) {
    *outText = nullptr;
    *outLen = 0;
    hgx_alignment *al = als[0];
    if (bedType > 9 || bedType == 7 || bedType < 0)
        return false;
    const GenomeTables &S = al->img.genomes[(size_t)srcGenome], &T = al->img.genomes[(size_t)tgtGenome];
    std::unordered_map<std::string, int> seqByName;
    for (size_t i = 0; i < S.seqs.size(); ++i)
        seqByName.emplace(S.seqs[i].name, (int)i);
    const bool timing = getenv("HGX_TEXT_TIMING") != nullptr;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    const unsigned hw = std::max(1u, hostBurstThreads()); // (three phases of a few milliseconds a call)
    static const unsigned maxThreads = getenv("HGX_TEXT_THREADS") ? (unsigned)std::max(1, atoi(getenv("HGX_TEXT_THREADS"))) : 64u;
    const unsigned threads = (unsigned)std::min<size_t>(std::min(hw, maxThreads), len / (1u << 18) + 1);
    // chunks: about four per thread, cut behind a newline; no chunk larger than 16 MB of text, so that a device batch (whole
    // chunks, at most batchLines intervals: below) stays bounded on inputs of any size
    std::vector<Chunk> chunks;
    {
        const size_t want = std::max<size_t>((size_t)threads * 4, len / ((size_t)16 << 20) + 1), per = len / want + 1;
        const char *p = text, *end = text + len;
        while (p < end) {
            const char *q = p + per < end ? p + per : end;
            if (q < end) {
                const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q));
                q = nl ? nl + 1 : end;
            }
            Chunk c;
            c.begin = p;
            c.end = q;
            chunks.push_back(std::move(c));
            p = q;
        }
    }
    forEachChunk(chunks, threads, [&](Chunk &C) { parseChunk(C, bedType, seqByName, S); });
    const auto t1 = now();
    // one column count for the whole input, nothing for the general path; the first malformed line ends the input
    int bt = -1;
    size_t usable = chunks.size();
    for (size_t i = 0; i < chunks.size(); ++i) {
        Chunk &C = chunks[i];
        if (C.general)
            return false;
        if (C.bedType >= 0) {
            if (bt >= 0 && C.bedType != bt)
                return false;
            bt = C.bedType;
        }
        if (!C.error.empty()) {
            usable = i + 1;
            break;
        }
    }
    size_t nq = 0, nlines = 0;
    for (size_t i = 0; i < usable; ++i) {
        Chunk &C = chunks[i];
        C.bedType = bt;
        C.firstLine = nlines;
        nq += C.numQueries;
        nlines += C.linesSeen;
        for (const std::string &n : C.notes) { // (stderr, like the reference; unknown sequences once per name)
            if (n[0] == '?') {
                if (missedSet.insert(n.substr(1)).second)
                    std::cerr << "Unable to find sequence " << n.substr(1) << " in genome " << S.name << std::endl;
            } else {
                std::cerr << n << std::endl;
            }
        }
        if (!C.error.empty())
            error = C.error + " in input bed line " + std::to_string(C.firstLine + C.errorLine);
    }
    chunks.resize(usable);
    stats = hgx_liftover_stats{};
    if (nq == 0)
        return true;
    // contiguous shares of the chunks for the devices, about the same number of intervals each
    std::vector<size_t> devQueries((size_t)nAls, 0);
    {
        const size_t share = (nq + (size_t)nAls - 1) / (size_t)nAls;
        int d = 0;
        for (Chunk &C : chunks) {
            if (devQueries[(size_t)d] >= share && d + 1 < nAls)
                ++d;
            C.device = d;
            C.firstQuery = devQueries[(size_t)d];
            devQueries[(size_t)d] += C.numQueries;
        }
    }
    // Every device works through its share in groups of whole chunks of at most batchLines intervals (a chunk of more than
    // that is a group of its own): stage, lift, render, next group — the plan, the pinned staging and the records in flight are
    // sized by a group, not by the input.  Round r runs the r-th group of every device at the same time.
    batchLines = std::max<size_t>(batchLines, 1);
    struct Group {
        size_t firstChunk, endChunk, numQueries;
    };
    std::vector<std::vector<Group>> groups((size_t)nAls);
    for (size_t i = 0; i < chunks.size();) {
        const int d = chunks[i].device;
        Group g{i, i, 0};
        while (g.endChunk < chunks.size() && chunks[g.endChunk].device == d &&
               (g.endChunk == g.firstChunk || g.numQueries + chunks[g.endChunk].numQueries <= batchLines)) {
            chunks[g.endChunk].firstQuery = g.numQueries; // (inside the group's batch)
            g.numQueries += chunks[g.endChunk].numQueries;
            ++g.endChunk;
        }
        groups[(size_t)d].push_back(g);
        i = g.endChunk;
    }
    size_t rounds = 0;
    for (int d = 0; d < nAls; ++d)
        rounds = std::max(rounds, groups[(size_t)d].size());
    hgx_liftover_opts opts{};
    opts.traverse_dupes = traverseDupes ? 1 : 0;
    opts.coalescence_limit = coalescenceLimit;
    double msStage = 0, msDevice = 0, msRender = 0;
    // a device error ends the input at the first line of the group that failed (below): groups behind it are not run, groups
    // in front of it — other devices' later rounds — still are
    size_t limit = chunks.size();
    std::string devFailure;
    const bool direct = nAls == 1 || rounds == 1;
    struct Text { // the output text while it grows (handed to the caller at the end; released if something throws)
        char *p = nullptr;
        ~Text() { textFree(p); }
    } lifted;
    size_t total = 0;
    for (size_t round = 0; round < rounds; ++round) {
        const auto r0 = now();
        std::vector<int64_t *> gs((size_t)nAls, nullptr), ge((size_t)nAls, nullptr);
        std::vector<uint8_t *> st((size_t)nAls, nullptr);
        std::vector<Chunk *> mine; // the chunks of this round
        std::vector<const Group *> grp((size_t)nAls, nullptr);
        for (int d = 0; d < nAls; ++d)
            if (round < groups[(size_t)d].size() && groups[(size_t)d][round].firstChunk < limit) {
                grp[(size_t)d] = &groups[(size_t)d][round];
#ifdef HGX_HOST_PROFILE
                static std::vector<std::vector<int64_t>> replayStarts, replayEnds; // (the profiling build's stand-ins for the plans' pinned staging)
                static std::vector<std::vector<uint8_t>> replayStrands;
                if (liftReplay().f) {
                    if (replayStarts.size() < (size_t)nAls) {
                        replayStarts.resize((size_t)nAls);
                        replayEnds.resize((size_t)nAls);
                        replayStrands.resize((size_t)nAls);
                    }
                    replayStarts[(size_t)d].resize(grp[(size_t)d]->numQueries + 1);
                    replayEnds[(size_t)d].resize(grp[(size_t)d]->numQueries + 1);
                    replayStrands[(size_t)d].resize(grp[(size_t)d]->numQueries + 1);
                    gs[(size_t)d] = replayStarts[(size_t)d].data();
                    ge[(size_t)d] = replayEnds[(size_t)d].data();
                    st[(size_t)d] = replayStrands[(size_t)d].data();
                } else
#endif
                if (grp[(size_t)d]->numQueries)
                    liftoverStageQueries(als[d], grp[(size_t)d]->numQueries, &gs[(size_t)d], &ge[(size_t)d], &st[(size_t)d]);
                for (size_t i = grp[(size_t)d]->firstChunk; i < grp[(size_t)d]->endChunk; ++i)
                    mine.push_back(&chunks[i]);
            }
        auto forMine = [&](auto f) {
            std::atomic<size_t> next{0};
            const std::function<void()> work = [&]() {
                for (size_t i; (i = next.fetch_add(1)) < mine.size();)
                    f(*mine[i]);
            };
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, mine.size()));
            if (textPool().run(nt, work))
                return;
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nt; ++t)
                pool.emplace_back(work);
            work();
            for (std::thread &t : pool)
                t.join();
        };
        forMine([&](Chunk &C) {
            size_t q = C.firstQuery;
            int64_t *s0 = gs[(size_t)C.device], *e0 = ge[(size_t)C.device];
            uint8_t *t0 = st[(size_t)C.device];
            for (Line &L : C.lines) {
                if (L.seq < 0) {
                    L.query = -1;
                    continue;
                }
                const int64_t off = S.seqs[(size_t)L.seq].start;
                s0[q] = L.start + off;     // halBlockLiftover.cpp:48
                e0[q] = L.end - 1 + off;   // :49
                t0[q] = (uint8_t)L.strand;
                L.query = (int64_t)q++;
            }
        });
        const auto r1 = now();
        std::vector<const hgx_record *> recs((size_t)nAls, nullptr);
        std::vector<PackedRecords> packedRecs((size_t)nAls);
        std::vector<size_t> nRecs((size_t)nAls, 0);
        std::vector<hgx_liftover_stats> devStats((size_t)nAls);
        std::vector<std::string> devError((size_t)nAls);
        {
            auto run = [&](int d) {
                if (!grp[(size_t)d] || !grp[(size_t)d]->numQueries)
                    return;
                try {
#ifdef HGX_HOST_PROFILE
                    static std::vector<std::vector<hgx_record>> replayRecsOf(64);
                    static std::vector<uint32_t> replayWords, replayFirst;
                    if (liftReplay().f && nAls <= 64) {
                        std::vector<hgx_record> &replayRecs = replayRecsOf[(size_t)d];
                        int64_t before = 0; // the intervals of the chunks in front of the group, in input order
                        for (size_t i = 0; i < grp[(size_t)d]->firstChunk; ++i)
                            before += (int64_t)chunks[i].numQueries;
                        liftReplay().range(before, grp[(size_t)d]->numQueries, replayRecs);
                        recs[(size_t)d] = replayRecs.data();
                        nRecs[(size_t)d] = replayRecs.size();
                        // HGX_REPLAY_PACKED: the records in the 8-byte form the device hands over when they fit it (start, length |
                        // sequence << 22 | strand << 29, every interval's first record), so that this reading of them is covered too
                        bool fits = nAls == 1 && getenv("HGX_REPLAY_PACKED") != nullptr;
                        for (const hgx_record &r : replayRecs)
                            fits = fits && r.tgt_start >= 0 && r.tgt_start < ((int64_t)1 << 32) && r.tgt_end - r.tgt_start < (1 << 22) && r.tgt_seq < 128;
                        if (fits) {
                            replayWords.assign(2 * replayRecs.size(), 0);
                            replayFirst.assign(grp[0]->numQueries + 1, 0);
                            for (size_t i = 0; i < replayRecs.size(); ++i) {
                                const hgx_record &r = replayRecs[i];
                                replayWords[2 * i] = (uint32_t)r.tgt_start;
                                replayWords[2 * i + 1] = (uint32_t)(r.tgt_end - r.tgt_start) | ((uint32_t)r.tgt_seq << 22) |
                                                         ((r.strand == '+' ? 0u : r.strand == '-' ? 1u : 2u) << 29);
                                ++replayFirst[(size_t)r.query + 1];
                            }
                            for (size_t q = 0; q < grp[0]->numQueries; ++q)
                                replayFirst[q + 1] += replayFirst[q];
                            packedRecs[0].words = replayWords.data();
                            packedRecs[0].first = replayFirst.data();
                        }
                    } else
#endif
                    liftoverBatchStaged(als[d], srcGenome, tgtGenome, grp[(size_t)d]->numQueries, opts, &recs[(size_t)d], &nRecs[(size_t)d],
                                        &devStats[(size_t)d], &packedRecs[(size_t)d]);
                } catch (std::exception &e) {
                    devError[(size_t)d] = e.what();
                }
            };
            std::vector<std::thread> pool;
            for (int d = 1; d < nAls; ++d)
                pool.emplace_back(run, d);
            run(0);
            for (std::thread &t : pool)
                t.join();
        }
        for (int d = 0; d < nAls; ++d) {
            if (!devError[(size_t)d].empty()) {
                // (the reference's scanner adds the line to whatever visitLine throws: here the first line of the group that failed;
                // what lies in front of it in input order keeps its output, like the lines in front of a malformed one)
                if (grp[(size_t)d]->firstChunk < limit) {
                    limit = grp[(size_t)d]->firstChunk;
                    devFailure = devError[(size_t)d] + " in input bed line " + std::to_string(chunks[limit].firstLine + 1);
                }
                continue;
            }
            stats.queries += devStats[(size_t)d].queries;
            stats.records += devStats[(size_t)d].records;
            stats.mapped_pieces += devStats[(size_t)d].mapped_pieces;
            stats.top_derefs += devStats[(size_t)d].top_derefs;
            stats.bottom_derefs += devStats[(size_t)d].bottom_derefs;
            stats.deferred_queries += devStats[(size_t)d].deferred_queries;
            stats.general_queries += devStats[(size_t)d].general_queries;
            stats.total_ms = std::max(stats.total_ms, devStats[(size_t)d].total_ms);
            stats.walk_ms = std::max(stats.walk_ms, devStats[(size_t)d].walk_ms);
            stats.composed_kind = devStats[(size_t)d].composed_kind;
            stats.composed_records = devStats[(size_t)d].composed_records;
        }
        const auto r2 = now();
        // the chunks' lines: counted, then written — at their place in the output text where that is known by now (one device, or
        // one round: everything in front of a chunk has been counted), else into the chunk's own text, put together at the end
        forMine([&](Chunk &C) {
            const size_t d = (size_t)C.device;
            C.outBytes = C.numQueries && devError[d].empty() ? measureChunk(C, recs[d], nRecs[d], packedRecs[d], T) : 0;
        });
        if (direct) {
            size_t at = total;
            for (Chunk *C : mine) { // (in input order: a device's share of the chunks is contiguous, the devices' shares follow each other)
                C->outAt = at;
                at += C->outBytes;
            }
            char *q = (char *)textRealloc(lifted.p, at + 1);
            if (!q)
                throw std::runtime_error("out of memory");
            lifted.p = q;
            total = at;
        }
        forMine([&](Chunk &C) {
            const size_t d = (size_t)C.device;
            if (C.outBytes) {
                if (!direct)
                    C.out.resize(C.outBytes);
                renderChunk(C, recs[d], nRecs[d], packedRecs[d], T, direct ? lifted.p + C.outAt : &C.out[0], C.outBytes);
            }
            std::vector<Line>().swap(C.lines); // (the tokens of a rendered chunk are not needed any more)
        });
        const auto r3 = now();
        msStage += ms(r0, r1);
        msDevice += ms(r1, r2);
        msRender += ms(r2, r3);
    }
    if (!devFailure.empty()) {
        error = devFailure; // (it lies in front of any malformed line, which ended the input)
        if (direct)
            total = chunks[limit].outAt; // (the failed group's chunks are empty; nothing behind them counts)
        for (size_t i = limit; i < chunks.size(); ++i)
            chunks[i].out.clear();
    }
    const auto t4 = now();
    auto t5 = t4;
    if (!direct) {
        total = 0;
        for (Chunk &C : chunks) {
            C.outAt = total;
            total += C.out.size();
        }
        // the output buffer is handed to the caller as it is (hgx_textmem.hpp: a mapping advised as huge pages, or the block the last
        // call's text was released from), untouched until the chunks copy themselves in: the first touch of its pages is spread over
        // the threads as well
        lifted.p = (char *)textAlloc(total + 1);
        if (!lifted.p)
            throw std::runtime_error("out of memory");
        t5 = now();
        char *const dst = lifted.p;
        forEachChunk(chunks, threads, [&](Chunk &C) {
            if (!C.out.empty())
                memcpy(dst + C.outAt, C.out.data(), C.out.size());
        });
    } else if (!lifted.p) {
        lifted.p = (char *)textAlloc(1);
        if (!lifted.p)
            throw std::runtime_error("out of memory");
    }
    char *buf = lifted.p;
    lifted.p = nullptr;
    buf[total] = '\0';
    *outText = buf;
    *outLen = total;
    if (timing)
        std::cerr << "[hgx text] " << threads << " threads, " << nAls << " device(s), " << chunks.size() << " chunks, " << rounds << " round(s): tokenise "
                  << ms(t0, t1) << " ms, stage " << msStage << ", device (H2D, kernels, D2H) " << msDevice << ", render " << msRender
                  << ", allocate output " << ms(t4, t5) << ", gather " << ms(t5, now()) << std::endl;
    return true;
}


// What a writer does with the blobs of its group (hgx_liftover_gather_writers leaves them in its buffer): the BED lines of the
// intervals they hold, rendered as hgx_liftover_convert renders them.  text: the input lines the blobs' intervals came from, in
// order — blob i answers the next n_queries(i) lines that are intervals (lines of sequences the source genome does not have are
// not, as in the conversion itself).  The fast path's BED forms only (no blocks).
void liftoverRenderBlobs(hgx_alignment *al, int srcGenome, int tgtGenome, const char *text, size_t len, int bedType, const void *const *blobs,
                         const size_t *blobBytes, int nBlobs, char **outText, size_t *outLen) {
    *outText = nullptr;
    *outLen = 0;
    if (bedType > 9 || bedType == 7 || bedType < 0)
        throw std::runtime_error("hgx_liftover_render_blobs: BED3 to BED9 lines only");
    const GenomeTables &S = al->img.genomes[(size_t)srcGenome], &T = al->img.genomes[(size_t)tgtGenome];
    std::unordered_map<std::string, int> seqByName;
    for (size_t i = 0; i < S.seqs.size(); ++i)
        seqByName.emplace(S.seqs[i].name, (int)i);
    // the records of all blobs as rows, their query numbers counted over the whole text
    std::vector<hgx_record> recs;
    size_t nqBlobs = 0;
    for (int b = 0; b < nBlobs; ++b) {
        const unsigned char *p = static_cast<const unsigned char *>(blobs[b]);
        const size_t have = blobBytes[b];
        if (!p || have < 32 || memcmp(p, "HGXW", 4) != 0)
            throw std::runtime_error("hgx_liftover_render_blobs: slot " + std::to_string(b) + " does not hold a wire blob");
        uint32_t fmt;
        uint64_t nq, nrec;
        memcpy(&fmt, p + 4, 4);
        memcpy(&nq, p + 16, 8);
        memcpy(&nrec, p + 24, 8);
        const unsigned char *body = p + 32;
        if (fmt == 0)
            throw std::runtime_error("hgx_liftover_render_blobs: the rank of slot " + std::to_string(b) + " had no blob for the batch");
        const size_t cbytes = fmt == 8 || fmt == 12 ? (2 * (size_t)nq + 7) / 8 * 8 : 0;
        if (fmt != 8 && fmt != 12 && fmt != 20 && fmt != 40)
            throw std::runtime_error("hgx_liftover_render_blobs: unknown wire format " + std::to_string(fmt));
        if (have < 32 + cbytes + (size_t)fmt * (size_t)nrec)
            throw std::runtime_error("hgx_liftover_render_blobs: slot " + std::to_string(b) + " is shorter than its blob");
        const size_t base = recs.size();
        recs.resize(base + (size_t)nrec);
        if (fmt == 8 || fmt == 12) {
            const size_t words = fmt / 4;
            size_t at = 0;
            for (uint64_t q = 0; q < nq; ++q) {
                uint16_t c;
                memcpy(&c, body + 2 * q, 2);
                for (uint16_t k = 0; k < c; ++k, ++at) {
                    if (at >= nrec)
                        throw std::runtime_error("hgx_liftover_render_blobs: a blob's counts hold more records than the blob");
                    uint32_t w[3];
                    memcpy(w, body + cbytes + at * fmt, fmt);
                    const uint32_t tail = w[words - 1];
                    hgx_record &r = recs[base + at];
                    memset(&r, 0, sizeof r);
                    r.query = (int64_t)(nqBlobs + q);
                    r.tgt_start = (int64_t)w[0];
                    r.tgt_end = r.tgt_start + (int64_t)(tail & ((1u << 22) - 1u));
                    r.src_start = fmt == 12 ? (int64_t)w[1] : -1;
                    r.tgt_seq = (int32_t)((tail >> 22) & 127u);
                    r.strand = "+-."[(tail >> 29) & 3u];
                    r.tgt_reversed = (uint8_t)(tail >> 31);
                }
            }
            if (at != nrec)
                throw std::runtime_error("hgx_liftover_render_blobs: a blob's counts do not add up to its records");
        } else if (fmt == 20) {
            for (uint64_t i = 0; i < nrec; ++i) {
                int32_t w[5];
                memcpy(w, body + 20 * i, 20);
                hgx_record &r = recs[base + i];
                memset(&r, 0, sizeof r);
                r.query = (int64_t)nqBlobs + w[0];
                r.tgt_start = w[1];
                r.tgt_end = w[2];
                r.src_start = w[3];
                r.tgt_seq = (int32_t)(((uint32_t)w[4] >> 16) & 0xFFFFu);
                r.strand = (char)(((uint32_t)w[4] >> 8) & 0xFFu);
                r.tgt_reversed = (uint8_t)((uint32_t)w[4] & 0xFFu);
            }
        } else {
            memcpy(recs.data() + base, body, 40 * (size_t)nrec);
            for (uint64_t i = 0; i < nrec; ++i)
                recs[base + i].query += (int64_t)nqBlobs;
        }
        nqBlobs += (size_t)nq;
    }
    for (size_t i = 1; i < recs.size(); ++i)
        if (recs[i].query < recs[i - 1].query)
            throw std::runtime_error("hgx_liftover_render_blobs: a blob's records are not in the order of its intervals");
    for (const hgx_record &r : recs)
        if (r.tgt_seq < 0 || (size_t)r.tgt_seq >= T.seqs.size())
            throw std::runtime_error("hgx_liftover_render_blobs: a record names a sequence the target genome does not have");
    // the lines, parsed the way the conversion parses them
    const unsigned threads = (unsigned)std::min<size_t>(std::min(std::max(1u, hostThreads()), 64u), len / (1u << 18) + 1);
    std::vector<Chunk> chunks;
    {
        const size_t want = std::max<size_t>((size_t)threads * 4, len / ((size_t)16 << 20) + 1), per = len / want + 1;
        const char *p = text, *end = text + len;
        while (p < end) {
            const char *q = p + per < end ? p + per : end;
            if (q < end) {
                const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q));
                q = nl ? nl + 1 : end;
            }
            Chunk c;
            c.begin = p;
            c.end = q;
            chunks.push_back(std::move(c));
            p = q;
        }
    }
    forEachChunk(chunks, threads, [&](Chunk &C) { parseChunk(C, bedType, seqByName, S); });
    int bt = -1;
    size_t nq = 0;
    for (Chunk &C : chunks) {
        if (C.general)
            throw std::runtime_error("hgx_liftover_render_blobs: lines with blocks are not this entry point's");
        if (!C.error.empty())
            throw std::runtime_error(C.error);
        if (C.bedType >= 0) {
            if (bt >= 0 && C.bedType != bt)
                throw std::runtime_error("hgx_liftover_render_blobs: lines of different numbers of columns");
            bt = C.bedType;
        }
    }
    for (Chunk &C : chunks) {
        C.bedType = bt;
        C.firstQuery = nq;
        size_t q = nq;
        for (Line &L : C.lines)
            L.query = L.seq < 0 ? -1 : (int64_t)q++;
        nq += C.numQueries;
    }
    if (nq != nqBlobs)
        throw std::runtime_error("hgx_liftover_render_blobs: the text holds " + std::to_string(nq) + " intervals, the blobs answer " + std::to_string(nqBlobs));
    const PackedRecords none;
    size_t total = 0;
    forEachChunk(chunks, threads, [&](Chunk &C) { C.outBytes = measureChunk(C, recs.data(), recs.size(), none, T); });
    for (Chunk &C : chunks) {
        C.outAt = total;
        total += C.outBytes;
    }
    char *out = static_cast<char *>(textAlloc(std::max<size_t>(total, 1)));
    if (!out)
        throw std::bad_alloc();
    try {
        forEachChunk(chunks, threads, [&](Chunk &C) { renderChunk(C, recs.data(), recs.size(), none, T, out + C.outAt, C.outBytes); });
    } catch (...) {
        textFree(out);
        throw;
    }
    *outText = out;
    *outLen = total;
}

} // namespace hgx
