#include "hgx_textmem.hpp"
#include "hgx_liftover_host.hpp"
#include "hgx_lift_replay.hpp"
#include <algorithm>
#include <cerrno>
#include <charconv>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <exception>
#include <sstream>
#include <thread>
#include <unordered_map>

namespace hgx {

// api/impl/halCommon.cpp:28-42 (empty fields are kept, a trailing separator adds none)
static void chopString(const std::string &in, char sep, std::vector<std::string> &out) {
    out.clear();
    size_t start = 0, end;
    while ((end = in.find(sep, start)) != std::string::npos) {
        out.push_back(in.substr(start, end - start));
        start = end + 1;
    }
    if (start < in.length())
        out.push_back(in.substr(start));
}
// api/impl/halCommon.cpp:44-52 (operator>> semantics: leading blanks skipped, trailing junk ignored)
// strtoll has the same reading (blanks, optional sign, decimal digits, stop at the first other character; nothing read or
// out of range = failure) at a tenth of the cost of a stringstream per field.
static int64_t strToInt(const std::string &s) {
    const char *b = s.c_str();
    char *e = nullptr;
    errno = 0;
    const long long v = strtoll(b, &e, 10);
    if (e == b || errno == ERANGE)
        throw std::runtime_error("Error converting string to int: " + s);
    return (int64_t)v;
}

void BedLine::parse(const std::string &lineBuffer, int type) {
    bedType = type;
    std::vector<std::string> row;
    chopString(lineBuffer, '\t', row);
    if (row.size() < 3)
        throw std::runtime_error("Expected at least three columns in BED record: " + lineBuffer);
    if (bedType == 0)
        bedType = std::min(int(row.size()), 12);
    if ((size_t)bedType > row.size()) // (an explicit --bedType beyond the line's columns: the reference indexes past its row here)
        throw std::runtime_error("Expected at least " + std::to_string(bedType) + " columns in BED record: " + lineBuffer);
    chrName = row[0];
    start = strToInt(row[1]);
    end = strToInt(row[2]);
    if (start >= end)
        throw std::runtime_error("Error zero or negative length BED range: " + lineBuffer);
    if (bedType > 3)
        name = row[3];
    if (bedType > 4)
        score = strToInt(row[4]);
    if (bedType > 5) {
        strand = row[5][0];
        if (strand != '.' && strand != '+' && strand != '-')
            throw std::runtime_error("Strand character must be + or - or ." + lineBuffer);
    }
    if (bedType > 6)
        thickStart = strToInt(row[6]);
    if (bedType > 7)
        thickEnd = strToInt(row[7]);
    if (bedType > 8) {
        std::vector<std::string> rgb;
        chopString(row[8], ',', rgb);
        if (rgb.size() > 3 || rgb.size() == 0)
            throw std::runtime_error("Error parsing BED itemRGB: " + lineBuffer);
        itemR = strToInt(rgb[0]);
        itemG = itemB = itemR;
        if (rgb.size() > 1)
            itemG = strToInt(rgb[1]);
        if (rgb.size() == 3)
            itemB = strToInt(rgb[2]);
    }
    if (bedType > 9) {
        if (bedType < 12)
            throw std::runtime_error("Error parsing BED, insufficient columns for blocks: " + lineBuffer);
        const size_t numBlocks = (size_t)strToInt(row[9]);
        std::vector<std::string> sizes, starts;
        chopString(row[10], ',', sizes);
        if (sizes.size() != numBlocks)
            throw std::runtime_error("Error parsing BED blockSizes: " + lineBuffer);
        chopString(row[11], ',', starts);
        if (starts.size() != numBlocks)
            throw std::runtime_error("Error parsing BED blockStarts: " + lineBuffer);
        blocks.resize(numBlocks);
        for (size_t i = 0; i < numBlocks; ++i) {
            blocks[i].length = strToInt(sizes[i]);
            blocks[i].start = strToInt(starts[i]);
            if (start + blocks[i].start + blocks[i].length > end)
                throw std::runtime_error("Error BED block out of range: " + lineBuffer);
        }
    }
    extra.clear();
    for (size_t i = (size_t)bedType; i < row.size(); i++)
        extra.push_back(row[i]);
}

static inline void appendInt(std::string &buf, int64_t v) {
    char tmp[24];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
    buf.append(tmp, (size_t)(r.ptr - tmp));
}

// BedLine::write (halBedLine.cpp:104-151) into a text buffer.  The lifted form substitutes chromosome, range, strand and
// thick range for the line's own (what BlockLiftover::liftInterval and Liftover::cleanResults change) without copying the line.
void BedLine::append(std::string &buf, const std::string &chrom, int64_t s, int64_t e, char str, int64_t tStart, int64_t tEnd) const {
    buf += chrom;
    buf += '\t';
    appendInt(buf, s);
    buf += '\t';
    appendInt(buf, e);
    if (bedType > 3) {
        buf += '\t';
        buf += name;
    }
    if (bedType > 4) {
        buf += '\t';
        appendInt(buf, score);
    }
    if (bedType > 5) {
        buf += '\t';
        buf += str;
    }
    if (bedType > 6) {
        buf += '\t';
        appendInt(buf, tStart);
    }
    if (bedType > 7) {
        buf += '\t';
        appendInt(buf, tEnd);
    }
    if (bedType > 8) {
        buf += '\t';
        appendInt(buf, itemR);
        buf += ',';
        appendInt(buf, itemG);
        buf += ',';
        appendInt(buf, itemB);
    }
    if (bedType > 9) {
        buf += '\t';
        appendInt(buf, (int64_t)blocks.size());
        for (size_t i = 0; i < blocks.size(); ++i) {
            buf += (i == 0 ? '\t' : ',');
            appendInt(buf, blocks[i].length);
        }
        for (size_t i = 0; i < blocks.size(); ++i) {
            buf += (i == 0 ? '\t' : ',');
            appendInt(buf, blocks[i].start);
        }
    }
    for (const std::string &x : extra) {
        buf += '\t';
        buf += x;
    }
    buf += '\n';
}

void BedLine::write(std::ostream &os) const {
    std::string buf;
    append(buf, chrName, start, end, strand, thickStart, thickEnd);
    os.write(buf.data(), (std::streamsize)buf.size());
}

void BedLine::expandToBed12() {
    if (bedType <= 3)
        name = "";
    if (bedType <= 4)
        score = 0;
    if (bedType <= 5)
        strand = '+';
    if (bedType <= 6)
        thickStart = start;
    if (bedType <= 7)
        thickEnd = end;
    if (bedType <= 8)
        itemR = itemG = itemB = 0;
    if (bedType <= 9) {
        blocks.resize(1);
        blocks[0].start = 0;
        blocks[0].length = end - start;
    }
    bedType = 12;
}

void BedLine::writePSL(std::ostream &os, bool prefixWithName) const {
    if (!validatePSL())
        throw std::runtime_error("Internal error: PSL does not validate");
    const PSLInfo &p = psl[0];
    if (prefixWithName)
        os << name << '\t';
    os << p.matches << '\t' << p.misMatches << '\t' << p.repMatches << '\t' << p.nCount << '\t' << p.qNumInsert << '\t' << p.qBaseInsert
       << '\t' << p.tNumInsert << '\t' << p.tBaseInsert << '\t' << p.qStrand << strand << '\t' << p.qSeqName << '\t' << p.qSeqSize << '\t'
       << (srcStart - (int64_t)p.qChromOffset) << '\t' << (p.qEnd - p.qChromOffset) << '\t' << chrName << '\t' << p.tSeqSize << '\t'
       << start << '\t' << end << '\t' << blocks.size() << '\t';
    for (const BedBlock &b : blocks)
        os << b.length << ',';
    os << '\t';
    for (size_t i = 0; i < p.qBlockStarts.size(); ++i) {
        int64_t s = p.qBlockStarts[i] - (int64_t)p.qChromOffset;
        if (p.qStrand == '-')
            s = (int64_t)p.qSeqSize - s - blocks[i].length;
        os << s << ',';
    }
    os << '\t';
    for (const BedBlock &b : blocks) {
        int64_t s = b.start + start;
        if (strand == '-')
            s = (int64_t)p.tSeqSize - s - b.length;
        os << s << ',';
    }
    os << '\n';
}

bool BedLine::validatePSL() const {
    if (psl.size() != 1 || blocks.empty())
        return false;
    const PSLInfo &p = psl[0];
    if (blocks.size() != p.qBlockStarts.size())
        return false;
    uint64_t tot = 0;
    for (const BedBlock &b : blocks)
        tot += (uint64_t)b.length;
    if (tot != p.matches + p.misMatches + p.repMatches + p.nCount)
        return false;
    if (tot + p.qBaseInsert != p.qEnd - (uint64_t)srcStart)
        return false;
    if (tot + p.tBaseInsert != (uint64_t)end - (uint64_t)start)
        return false;
    if (strand != '-') {
        if (blocks[0].start != 0 || blocks.back().start + blocks.back().length + start != end)
            return false;
    } else {
        if (blocks.back().start != 0 || blocks[0].start + blocks[0].length + start != end)
            return false;
    }
    if (p.qStrand != '-') {
        if (p.qBlockStarts[0] != srcStart || (uint64_t)(p.qBlockStarts.back() + blocks.back().length) != p.qEnd)
            return false;
    } else {
        if (p.qBlockStarts.back() != srcStart || (uint64_t)(p.qBlockStarts[0] + blocks[0].length) != p.qEnd)
            return false;
    }
    return true;
}

// halLiftover.cpp:169-195
bool Liftover::compatible(const BedLine &tgtBed, const BedLine &newBlock) const {
    if (tgtBed.strand != newBlock.strand || tgtBed.srcStart == newBlock.srcStart)
        return false;
    const BedBlock &tb = tgtBed.blocks.back();
    const int64_t delta = tgtBed.strand != _inStrand ? tb.start - newBlock.end : newBlock.start - (tb.start + tb.length);
    return delta >= 0 && tgtBed.chrName == newBlock.chrName;
}

// halLiftover.cpp:197-234
void Liftover::flipBlocks(std::vector<BedLine> &lines) const {
    for (BedLine &b : lines) {
        if (b.blocks.size() > 1) {
            const int64_t delta = b.blocks[1].start - (b.blocks[0].start + b.blocks[0].length);
            const bool mustFlip = !_outPSL ? delta < 0 : ((b.strand == '-' && delta >= 0) || (b.strand != '-' && delta < 0));
            if (mustFlip) {
                std::reverse(b.blocks.begin(), b.blocks.end());
                if (_outPSL)
                    std::reverse(b.psl[0].qBlockStarts.begin(), b.psl[0].qBlockStarts.end());
            }
        }
    }
}

// halLiftover.cpp:236-290 (the iterator swaps of the reference written as index selection)
void Liftover::computePSLInserts(std::vector<BedLine> &lines) const {
    for (BedLine &bed : lines) {
        PSLInfo &p = bed.psl[0];
        p.qNumInsert = p.qBaseInsert = p.tNumInsert = p.tBaseInsert = 0;
        for (size_t i = 1; i < bed.blocks.size(); ++i) {
            // target gap: previous block in target order, then the next one
            const BedBlock &tA = bed.strand == '-' ? bed.blocks[i] : bed.blocks[i - 1];
            const BedBlock &tB = bed.strand == '-' ? bed.blocks[i - 1] : bed.blocks[i];
            uint64_t gap = (uint64_t)(tB.start - (tA.start + tA.length));
            if (gap > 0) {
                ++p.tNumInsert;
                p.tBaseInsert += gap;
            }
            // query gap
            const size_t a = p.qStrand == '-' ? i : i - 1, b = p.qStrand == '-' ? i - 1 : i;
            const int64_t qa = p.qBlockStarts[a], qb = p.qBlockStarts[b], la = bed.blocks[a].length;
            gap = qb >= qa + la ? (uint64_t)(qb - (qa + la)) : 0; // duplicated blocks can overlap
            if (gap > 0) {
                ++p.qNumInsert;
                p.qBaseInsert += gap;
            }
        }
    }
}

// halLiftover.cpp:108-167; mappedBlocks arrive stably sorted by source start
void Liftover::assignBlocksToIntervals(std::vector<BedLine> &mappedBlocks, std::vector<BedLine> &out) {
    int64_t prevSrcBlockEnd = NULL_INDEX;
    for (size_t k = 0; k < mappedBlocks.size(); ++k) {
        BedLine &blk = mappedBlocks[k];
        const int64_t srcBlockEnd = blk.srcStart + (blk.end - blk.start);
        const bool dupe = blk.srcStart < prevSrcBlockEnd || (k + 1 < mappedBlocks.size() && mappedBlocks[k + 1].srcStart < srcBlockEnd);
        if (out.empty() || (_outPSL && dupe) || !compatible(out.back(), blk))
            out.push_back(blk);
        prevSrcBlockEnd = srcBlockEnd;
        BedLine &tgt = out.back();
        tgt.start = std::min(tgt.start, blk.start);
        tgt.end = std::max(tgt.end, blk.end);
        tgt.blocks.push_back(BedBlock{blk.start, blk.end - blk.start}); // absolute for now
        if (_outPSL) {
            tgt.psl[0].qBlockStarts.push_back(blk.srcStart);
            if (tgt.blocks.size() > 1) {
                tgt.psl[0].matches += blk.psl[0].matches;
                tgt.psl[0].misMatches += blk.psl[0].misMatches;
                tgt.psl[0].repMatches += blk.psl[0].repMatches;
                tgt.psl[0].nCount += blk.psl[0].nCount;
            }
        }
    }
    for (BedLine &b : out)
        for (BedBlock &bb : b.blocks)
            bb.start -= b.start;
    if (!out.empty())
        flipBlocks(out);
    if (_outPSL)
        computePSLInserts(out);
}

// BlockLiftover::readPSLInfo (halBlockLiftover.cpp:115-162).  A lifted line is a run of fragments that are
// consecutive on both genomes (canMergeRightWith), so its base pairs follow one linear map: target offset d pairs
// with source offset d (same relative strand) or len-1-d (opposite); each side is complemented when its iterator
// is reversed.
static void pslCounts(const GenomeTables &S, const GenomeTables &T, int64_t sLo, int64_t tLo, int64_t len, bool srcRev, bool tgtRev,
                      PSLInfo &p) {
    if (S.dna.empty() || T.dna.empty())
        throw std::runtime_error("PSL output needs DNA, and this alignment image carries none");
    auto comp = [](char c) {
        switch (c) {
        case 'A': return 'T';
        case 'a': return 't';
        case 'C': return 'G';
        case 'c': return 'g';
        case 'G': return 'C';
        case 'g': return 'c';
        case 'T': return 'A';
        case 't': return 'a';
        default: return c;
        }
    };
    const bool opposite = srcRev != tgtRev;
    for (int64_t d = 0; d < len; ++d) {
        char sc = dnaAt(S.dna, opposite ? sLo + len - 1 - d : sLo + d), tc = dnaAt(T.dna, tLo + d);
        if (srcRev)
            sc = comp(sc);
        if (tgtRev)
            tc = comp(tc);
        if (sc == tc) {
            if (!(sc >= 'a') && !(tc >= 'a')) // isMasked (halCommon.h:130-132)
                ++p.matches;
            else
                ++p.repMatches;
        } else if (tc == 'n' || tc == 'N') { // isMissingData (:126-128)
            ++p.nCount;
        } else {
            ++p.misMatches;
        }
    }
}

void Liftover::convert(hgx_alignment *al, int srcGenome, std::istream *in, int tgtGenome, std::ostream *out, int bedType,
                       bool traverseDupes, bool outPSL, bool outPSLWithName, int coalescenceLimit) {
    if (in->bad())
        throw std::runtime_error("Error reading bed input stream");
    // the whole input in memory: the common shapes are parsed, lifted and printed by all cores at once
    std::string text((std::istreambuf_iterator<char>(*in)), std::istreambuf_iterator<char>());
    char *lifted = nullptr;
    size_t n = 0;
    try {
        convertBuffer(al, srcGenome, text.data(), text.size(), tgtGenome, &lifted, &n, bedType, traverseDupes, outPSL, outPSLWithName, coalescenceLimit);
    } catch (...) { // what was lifted before the failing line has been written by then in the reference, too
        if (lifted)
            out->write(lifted, (std::streamsize)n);
        textFree(lifted);
        throw;
    }
    if (lifted)
        out->write(lifted, (std::streamsize)n);
    textFree(lifted); // (hgx_textmem.hpp: the parallel path's text is a mapping, the general path's malloc's: it tells them apart)
}

void Liftover::convertBuffer(hgx_alignment *al, int srcGenome, const char *text, size_t len, int tgtGenome, char **outText, size_t *outLen, int bedType,
                             bool traverseDupes, bool outPSL, bool outPSLWithName, int coalescenceLimit) {
    *outText = nullptr;
    *outLen = 0;
    if (!(outPSL || outPSLWithName) && !getenv("HGX_TEXT_GENERAL")) {
        std::string error;
        _missedSet.clear();
        std::vector<hgx_alignment *> als{al};
        als.insert(als.end(), moreDevices.begin(), moreDevices.end());
        size_t lines = batchLines;
        if (const char *e = getenv("HGX_BATCH_LINES"))
            lines = (size_t)std::max<long long>(1, atoll(e));
        if (liftoverTextFast(als.data(), (int)als.size(), srcGenome, text, len, tgtGenome, bedType, traverseDupes, coalescenceLimit, outText, outLen,
                             error, _missedSet, lastStats, lines)) {
            if (!error.empty())
                throw std::runtime_error(error);
            return;
        }
    }
    std::istringstream is(std::string(text, len));
    std::ostringstream os;
    auto hand = [&]() {
        const std::string s = os.str();
        *outText = (char *)malloc(s.size() + 1);
        if (!*outText)
            throw std::runtime_error("out of memory");
        memcpy(*outText, s.c_str(), s.size() + 1);
        *outLen = s.size();
    };
    try {
        convertGeneral(al, srcGenome, &is, tgtGenome, &os, bedType, traverseDupes, outPSL, outPSLWithName, coalescenceLimit);
    } catch (...) {
        hand();
        throw;
    }
    hand();
}

void Liftover::convertGeneral(hgx_alignment *al, int srcGenome, std::istream *in, int tgtGenome, std::ostream *out, int bedType,
                              bool traverseDupes, bool outPSL, bool outPSLWithName, int coalescenceLimit) {
    if (const char *e = getenv("HGX_BATCH_LINES")) // (intervals per device batch: tests cross batch ends with it)
        batchLines = (size_t)std::max<long long>(1, atoll(e));
    _outPSL = outPSL || outPSLWithName; // halLiftoverMain.cpp:82-84
    _outPSLWithName = outPSLWithName;
    const GenomeTables &S = al->img.genomes[(size_t)srcGenome];
    const GenomeTables &T = al->img.genomes[(size_t)tgtGenome];
    std::unordered_map<std::string, int> seqByName;
    for (size_t i = 0; i < S.seqs.size(); ++i)
        seqByName.emplace(S.seqs[i].name, (int)i);
    hgx_liftover_opts opts{};
    opts.traverse_dupes = traverseDupes ? 1 : 0;
    opts.coalescence_limit = coalescenceLimit;
    opts.min_length = 0;
    _missedSet.clear();
    lastStats = hgx_liftover_stats{};

    if (in->bad())
        throw std::runtime_error("Error reading bed input stream");
    // The scanner's loop (halBedScanner.cpp:40-61, 76-80: blank space skipped, a line read, parsed, visited) runs over the text in
    // pieces cut behind line ends, a piece a thread: a line's BedLine depends on the lines before it only through the fields a shorter
    // line leaves as they were (BedScanner keeps one BedLine) — the strand a line of five columns or fewer is lifted on, the thick
    // end that decides for a line of seven columns whether its thick start is the lifted one.  So every piece is parsed from a fresh
    // BedLine, and one pass in order gives every line the fields it has no column for from the line before it (as they stand after
    // that line), counts the lines, reports the skipped ones and stops at the first malformed one.
    const std::string text((std::istreambuf_iterator<char>(*in)), std::istreambuf_iterator<char>());
    struct Item {
        BedLine line;
        int seq = -1;          // source sequence; -1: the line is skipped
        int columns = 0;       // the line's own column count (BedLine::bedType as parse left it): the fields behind are the scanner's
        int note = 0;          // skipped: 1 unknown sequence, 2 past the sequence's end, 3 no blocks
    };
    struct Piece {
        size_t begin = 0, end = 0;
        std::vector<Item> items;
        std::string error; // the first malformed line's message (the piece's items end before it)
    };
    auto parsePiece = [&](Piece &pc) {
        BedLine bedLine;
        std::string lineBuffer;
        size_t pos = pc.begin;
        auto skipWhiteSpaces = [&]() {
            while (pos < pc.end && std::isspace((unsigned char)text[pos]))
                ++pos;
        };
        skipWhiteSpaces();
        while (pos < pc.end) {
            const char *nl = (const char *)memchr(text.data() + pos, '\n', pc.end - pos);
            const size_t eol = nl ? (size_t)(nl - text.data()) : pc.end;
            lineBuffer.assign(text, pos, eol - pos);
            pos = nl ? eol + 1 : pc.end;
            try {
                bedLine.parse(lineBuffer, bedType);
            } catch (std::runtime_error &e) {
                pc.error = e.what();
                return;
            }
            pc.items.emplace_back();
            Item &it = pc.items.back();
            it.columns = bedLine.bedType;
            // Liftover::visitLine, halLiftover.cpp:46-70
            if (_outPSL && bedLine.bedType < 12)
                bedLine.expandToBed12(); // forcing to BED12 makes PSL code simpler
            auto found = seqByName.find(bedLine.chrName);
            if (found == seqByName.end()) {
                it.note = 1;
            } else if (bedLine.end > S.seqs[(size_t)found->second].length) {
                it.note = 2;
                it.seq = found->second; // (for the message; the line is skipped all the same)
            } else if (bedLine.bedType > 9 && bedLine.blocks.empty()) {
                it.note = 3;
            } else {
                it.seq = found->second;
                if (bedLine.bedType > 9) // liftBlockIntervals, halLiftover.cpp:296-309: blocks sorted by start
                    std::sort(bedLine.blocks.begin(), bedLine.blocks.end());
            }
            it.line = bedLine;
            skipWhiteSpaces();
        }
    };
    size_t lineNumber = 0;
    BedLine before; // the scanner's BedLine as the lines so far left it (the fields a shorter line keeps)
    std::string pendingError;
    struct Job {
        const BedLine *line;
        size_t firstQuery, numQueries;
    };
    std::vector<Job> jobs;
    std::vector<hgx_interval> ivs;
    std::vector<hgx_record> recs;
    size_t batchFirstLine = 1;
    auto liftBatch = [&]() {
        if (!jobs.empty()) {
            hgx_liftover_stats st{};
            try { // (BedScanner::scan wraps visitLine as well, halBedScanner.cpp:60-68: an error raised while lifting names a line — here
                  // the first line of the batch it was lifted in)
#ifdef HGX_HOST_PROFILE
                if (liftReplay().f)
                    liftReplay().batch(ivs.size(), recs);
                else
#endif
                    liftoverBatchHost(al, srcGenome, tgtGenome, ivs.size(), ivs.data(), opts, recs, &st);
            } catch (std::runtime_error &e) {
                throw std::runtime_error(std::string(e.what()) + " in input bed line " + std::to_string(batchFirstLine));
            }
            lastStats.queries += st.queries;
            lastStats.source_pieces += st.source_pieces;
            lastStats.top_derefs += st.top_derefs;
            lastStats.bottom_derefs += st.bottom_derefs;
            lastStats.mapped_pieces += st.mapped_pieces;
            lastStats.records += st.records;
            lastStats.deferred_queries += st.deferred_queries;
            lastStats.walk_ms += st.walk_ms;
            lastStats.total_ms += st.total_ms;
            // The lines' records (grouped by query, in input order), then the lines themselves: what happens to a line's records —
            // BlockLiftover's output lines, assignBlocksToIntervals, cleanResults, the PSL columns — depends on that line only, so
            // the batch's lines are dealt to the host's threads in contiguous shares and the shares' texts written in order.
            std::vector<size_t> jobRec(jobs.size() + 1);
            {
                size_t r = 0;
                for (size_t j = 0; j < jobs.size(); ++j) {
                    jobRec[j] = r;
                    const size_t qEnd = jobs[j].firstQuery + jobs[j].numQueries;
                    while (r < recs.size() && (size_t)recs[r].query < qEnd)
                        ++r;
                }
                jobRec[jobs.size()] = r;
            }
            struct Share {
                std::string text;
                std::exception_ptr error; // (then `text` ends where the serial loop would have stopped writing)
            };
            unsigned nt = hostBurstThreads();
            nt = std::max(1u, std::min(nt ? nt : 1u, 32u));
            nt = (unsigned)std::min<size_t>(nt, jobs.size() / 64 + 1);
            std::vector<Share> shares(nt);
            const bool outPSL = _outPSL, outPSLWithName = _outPSLWithName;
            auto render = [&](unsigned t) {
                Share &share = shares[t];
                std::string &outBuf = share.text;
                Liftover self; // (assignBlocksToIntervals and its helpers read the input line's strand from the object)
                self._outPSL = outPSL;
                self._outPSLWithName = outPSLWithName;
                std::ostringstream pslText;
                std::vector<BedLine> mapped, outLines;
                try {
                    for (size_t j = jobs.size() * t / nt; j < jobs.size() * (t + 1) / nt; ++j) {
                        const BedLine &src = *jobs[j].line;
                        const size_t r0 = jobRec[j], r = jobRec[j + 1];
                        if (src.bedType <= 9) {
                            for (size_t k = r0; k < r; ++k) {
                                const hgx_record &rec = recs[k];
                                // halBlockLiftover.cpp:82-105 (fields other than chrom/start/end/strand echo the input line);
                                // Liftover::cleanResults, halLiftover.cpp:313-331: a set thick range becomes the lifted range
                                const bool thick = src.bedType > 6 && (src.thickStart != 0 || src.thickEnd != 0);
                                src.append(outBuf, T.seqs[(size_t)rec.tgt_seq].name, rec.tgt_start, rec.tgt_end, rec.strand,
                                           thick ? rec.tgt_start : src.thickStart, thick ? rec.tgt_end : src.thickEnd);
                            }
                            continue;
                        }
                        // BED12 / PSL: mapped blocks of all of the line's block intervals, stably sorted by source start
                        // (assignBlocksToIntervals' first step; per interval they already are)
                        self._inStrand = src.strand;
                        mapped.clear();
                        outLines.clear();
                        for (size_t k = r0; k < r; ++k) {
                            const hgx_record &rec = recs[k];
                            mapped.push_back(src);
                            BedLine &m = mapped.back();
                            m.blocks.clear();
                            m.chrName = T.seqs[(size_t)rec.tgt_seq].name;
                            m.start = rec.tgt_start;
                            m.end = rec.tgt_end;
                            m.strand = rec.strand;
                            m.srcStart = rec.src_start;
                            if (outPSL) {
                                const SeqInfo &qs = S.seqs[(size_t)ivs[(size_t)rec.query].seq];
                                const SeqInfo &ts = T.seqs[(size_t)rec.tgt_seq];
                                m.psl.assign(1, PSLInfo());
                                PSLInfo &p = m.psl[0];
                                p.qSeqName = qs.name;
                                p.qSeqSize = (uint64_t)qs.length;
                                p.qStrand = src.strand == '-' ? '-' : '+'; // source pieces are flipped for '-' input (halBlockLiftover.cpp:64-70)
                                p.qChromOffset = (uint64_t)qs.start;
                                p.qEnd = (uint64_t)(m.srcStart + (m.end - m.start));
                                p.tSeqSize = (uint64_t)ts.length;
                                pslCounts(S, T, m.srcStart, m.start + ts.start, m.end - m.start, src.strand == '-', rec.tgt_reversed != 0, p);
                            }
                        }
                        std::stable_sort(mapped.begin(), mapped.end(), [](const BedLine &a, const BedLine &b) { return a.srcStart < b.srcStart; });
                        if (!mapped.empty())
                            self.assignBlocksToIntervals(mapped, outLines);
                        // cleanResults (halLiftover.cpp:313-355)
                        for (BedLine &b : outLines) {
                            if (src.thickStart != 0 || src.thickEnd != 0) {
                                b.thickStart = b.start;
                                b.thickEnd = b.end;
                            }
                            if (outPSL) {
                                b.srcStart = INT64_MAX;
                                b.psl[0].qEnd = 0;
                                for (size_t k = 0; k < b.psl[0].qBlockStarts.size(); ++k) {
                                    b.srcStart = std::min(b.srcStart, b.psl[0].qBlockStarts[k]);
                                    b.psl[0].qEnd = std::max(b.psl[0].qEnd, (uint64_t)b.psl[0].qBlockStarts[k] + (uint64_t)b.blocks[k].length);
                                }
                            }
                        }
                        std::stable_sort(outLines.begin(), outLines.end(), [](const BedLine &a, const BedLine &b) { return a.srcStart < b.srcStart; });
                        for (const BedLine &b : outLines) {
                            if (!outPSL) {
                                b.append(outBuf, b.chrName, b.start, b.end, b.strand, b.thickStart, b.thickEnd); // (BedLine::write)
                            } else {
                                pslText.str(std::string());
                                b.writePSL(pslText, outPSLWithName);
                                outBuf += pslText.str();
                            }
                        }
                    }
                } catch (...) {
                    share.error = std::current_exception();
                }
            };
            if (nt == 1) {
                render(0);
            } else {
                std::vector<std::thread> threads;
                for (unsigned t = 1; t < nt; ++t)
                    threads.emplace_back(render, t);
                render(0);
                for (std::thread &th : threads)
                    th.join();
            }
            for (Share &share : shares) { // (what was written before a line that fails stays written, as in the reference's loop)
                if (!share.text.empty())
                    out->write(share.text.data(), (std::streamsize)share.text.size());
                if (share.error)
                    std::rethrow_exception(share.error);
            }
        }
        jobs.clear();
        ivs.clear();
    };
    const size_t window = (size_t)64 << 20; // (of text at a time: its lines' BedLines are all in memory)
    for (size_t w0 = 0; w0 < text.size() && pendingError.empty();) {
        size_t w1 = std::min(text.size(), w0 + window);
        if (w1 < text.size()) {
            const char *nl = (const char *)memchr(text.data() + w1, '\n', text.size() - w1);
            w1 = nl ? (size_t)(nl - text.data()) + 1 : text.size();
        }
        unsigned np = hostBurstThreads();
        np = std::max(1u, std::min(np ? np : 1u, 32u));
        static const size_t pieceBytes = getenv("HGX_PARSE_PIECE") ? (size_t)std::max(1, atoi(getenv("HGX_PARSE_PIECE"))) : 16384; // (tests: small pieces)
        np = (unsigned)std::min<size_t>(np, (w1 - w0) / pieceBytes + 1);
        std::vector<Piece> pieces(np);
        for (unsigned k = 0; k < np; ++k) { // cut behind line ends
            pieces[k].begin = k == 0 ? w0 : pieces[k - 1].end;
            size_t e = k + 1 == np ? w1 : std::max(pieces[k].begin, w0 + (w1 - w0) * (k + 1) / np);
            if (e < w1) {
                const char *nl = (const char *)memchr(text.data() + e, '\n', w1 - e);
                e = nl ? (size_t)(nl - text.data()) + 1 : w1;
            }
            pieces[k].end = e;
        }
        if (np == 1) {
            parsePiece(pieces[0]);
        } else {
            std::vector<std::thread> threads;
            for (unsigned k = 1; k < np; ++k)
                threads.emplace_back([&, k]() { parsePiece(pieces[k]); });
            parsePiece(pieces[0]);
            for (std::thread &th : threads)
                th.join();
        }
        // in order: line numbers, the strand handed on, the skipped lines' messages, batches of intervals, the first malformed line
        for (Piece &pc : pieces) {
            for (Item &it : pc.items) {
                ++lineNumber;
                BedLine &bl = it.line;
                if (!_outPSL) { // (with PSL output expandToBed12 has given the missing fields their defaults, in the scanner's BedLine too)
                    if (it.columns <= 3)
                        bl.name = before.name;
                    if (it.columns <= 4)
                        bl.score = before.score;
                    if (it.columns <= 5)
                        bl.strand = before.strand; // (such a line is lifted on the strand the scanner's BedLine still holds)
                    if (it.columns <= 6)
                        bl.thickStart = before.thickStart;
                    if (it.columns <= 7)
                        bl.thickEnd = before.thickEnd; // (a line of seven columns: a thick end left over decides whether its thick start is lifted)
                    if (it.columns <= 8) {
                        bl.itemR = before.itemR;
                        bl.itemG = before.itemG;
                        bl.itemB = before.itemB;
                    }
                }
                before.name = bl.name;
                before.score = bl.score;
                before.strand = bl.strand;
                before.thickStart = bl.thickStart;
                before.thickEnd = bl.thickEnd;
                before.itemR = bl.itemR;
                before.itemG = bl.itemG;
                before.itemB = bl.itemB;
                if (it.note == 1) {
                    if (_missedSet.insert(bl.chrName).second)
                        std::cerr << "Unable to find sequence " << bl.chrName << " in genome " << S.name << std::endl;
                } else if (it.note == 2) {
                    std::cerr << "Skipping interval with endpoint " << bl.end << "because sequence " << bl.chrName << " has length "
                              << S.seqs[(size_t)it.seq].length << std::endl;
                } else if (it.note == 3) {
                    std::cerr << "Skipping input line with 0 blocks" << std::endl;
                } else {
                    if (jobs.empty())
                        batchFirstLine = lineNumber;
                    Job job;
                    job.line = &bl;
                    job.firstQuery = ivs.size();
                    hgx_interval q;
                    q.seq = it.seq;
                    q.strand = bl.strand;
                    q._pad[0] = q._pad[1] = q._pad[2] = 0;
                    if (bl.bedType <= 9) {
                        q.start = bl.start;
                        q.end = bl.end;
                        ivs.push_back(q);
                    } else { // one lift per non-empty block
                        for (const BedBlock &b : bl.blocks) {
                            q.start = b.start + bl.start;
                            q.end = q.start + b.length;
                            if (q.end > q.start)
                                ivs.push_back(q);
                        }
                    }
                    job.numQueries = ivs.size() - job.firstQuery;
                    jobs.push_back(job);
                    if (ivs.size() >= batchLines)
                        liftBatch();
                }
            }
            if (!pc.error.empty()) {
                pendingError = pc.error + " in input bed line " + std::to_string(lineNumber + 1);
                break;
            }
        }
        liftBatch(); // (the window's BedLines go with it)
        w0 = w1;
    }
    if (!pendingError.empty())
        throw std::runtime_error(pendingError);
}

} // namespace hgx
