#include "hgx_liftover_host.hpp"
#include <iostream>
#include <sstream>
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
static int64_t strToInt(const std::string &s) {
    std::stringstream ss(s);
    int64_t i;
    ss >> i;
    if (ss.bad() || ss.fail())
        throw std::runtime_error("Error converting string to int: " + s);
    return i;
}

void BedLine::parse(const std::string &lineBuffer, int type) {
    bedType = type;
    std::vector<std::string> row;
    chopString(lineBuffer, '\t', row);
    if (row.size() < 3)
        throw std::runtime_error("Expected at least three columns in BED record: " + lineBuffer);
    if (bedType == 0)
        bedType = std::min(int(row.size()), 12);
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
    if (bedType > 9)
        throw std::runtime_error("BED12 block lifting (and PSL output) is not built yet; use --bedType 9 or fewer columns: " +
                                 lineBuffer);
    extra.clear();
    for (size_t i = (size_t)bedType; i < row.size(); i++)
        extra.push_back(row[i]);
}

void BedLine::write(std::ostream &os) const {
    os << chrName << '\t' << start << '\t' << end;
    if (bedType > 3)
        os << '\t' << name;
    if (bedType > 4)
        os << '\t' << score;
    if (bedType > 5)
        os << '\t' << strand;
    if (bedType > 6)
        os << '\t' << thickStart;
    if (bedType > 7)
        os << '\t' << thickEnd;
    if (bedType > 8)
        os << '\t' << itemR << ',' << itemG << ',' << itemB;
    for (const std::string &e : extra)
        os << '\t' << e;
    os << '\n';
}

void Liftover::convert(hgx_alignment *al, int srcGenome, std::istream *in, int tgtGenome, std::ostream *out, int bedType,
                       bool traverseDupes, bool outPSL, bool outPSLWithName, int coalescenceLimit) {
    if (outPSL || outPSLWithName)
        throw std::runtime_error("PSL output is not built yet (SURVEY 8(f) item 1)");
    const GenomeTables &S = al->img.genomes[(size_t)srcGenome];
    const GenomeTables &T = al->img.genomes[(size_t)tgtGenome];
    std::unordered_map<std::string, int> seqByName;
    for (size_t i = 0; i < S.seqs.size(); ++i)
        seqByName.emplace(S.seqs[i].name, (int)i);
    hgx_liftover_opts opts;
    opts.traverse_dupes = traverseDupes ? 1 : 0;
    opts.coalescence_limit = coalescenceLimit;
    opts.min_length = 0;
    _missedSet.clear();
    lastStats = hgx_liftover_stats{};

    if (in->bad())
        throw std::runtime_error("Error reading bed input stream");
    auto skipWhiteSpaces = [](std::istream *s) { // halBedScanner.cpp:76-80
        while (s->good() && std::isspace((char)s->peek()))
            s->get();
    };
    BedLine bedLine; // persists across lines like BedScanner::_bedLine (fields of shorter lines are inherited)
    std::string lineBuffer;
    size_t lineNumber = 0;
    std::string pendingError;
    std::vector<BedLine> lines;
    std::vector<hgx_interval> ivs;
    std::vector<hgx_record> recs;
    skipWhiteSpaces(in);
    bool more = in->good();
    while (more || !pendingError.empty()) {
        lines.clear();
        ivs.clear();
        while (more && lines.size() < batchLines) {
            ++lineNumber;
            try {
                std::getline(*in, lineBuffer);
                bedLine.parse(lineBuffer, bedType);
            } catch (std::runtime_error &e) {
                pendingError = std::string(e.what()) + " in input bed line " + std::to_string(lineNumber);
                more = false;
                break;
            }
            // Liftover::visitLine, halLiftover.cpp:51-66
            auto it = seqByName.find(bedLine.chrName);
            if (it == seqByName.end()) {
                if (_missedSet.insert(bedLine.chrName).second)
                    std::cerr << "Unable to find sequence " << bedLine.chrName << " in genome " << S.name << std::endl;
            } else if (bedLine.end > S.seqs[(size_t)it->second].length) {
                std::cerr << "Skipping interval with endpoint " << bedLine.end << "because sequence " << bedLine.chrName
                          << " has length " << S.seqs[(size_t)it->second].length << std::endl;
            } else {
                hgx_interval q;
                q.start = bedLine.start;
                q.end = bedLine.end;
                q.seq = it->second;
                q.strand = bedLine.strand;
                q._pad[0] = q._pad[1] = q._pad[2] = 0;
                ivs.push_back(q);
                lines.push_back(bedLine);
            }
            skipWhiteSpaces(in);
            more = in->good();
        }
        if (!ivs.empty()) {
            hgx_liftover_stats st{};
            liftoverBatchHost(al, srcGenome, tgtGenome, ivs.size(), ivs.data(), opts, recs, &st);
            lastStats.queries += st.queries;
            lastStats.source_pieces += st.source_pieces;
            lastStats.top_derefs += st.top_derefs;
            lastStats.bottom_derefs += st.bottom_derefs;
            lastStats.mapped_pieces += st.mapped_pieces;
            lastStats.records += st.records;
            lastStats.deferred_queries += st.deferred_queries;
            lastStats.walk_ms += st.walk_ms;
            lastStats.total_ms += st.total_ms;
            BedLine o;
            for (const hgx_record &r : recs) {
                // halBlockLiftover.cpp:82-105 (fields other than chrom/start/end/strand echo the input line)
                o = lines[(size_t)r.query];
                o.chrName = T.seqs[(size_t)r.tgt_seq].name;
                o.start = r.tgt_start;
                o.end = r.tgt_end;
                o.strand = r.strand;
                // Liftover::cleanResults, halLiftover.cpp:313-331
                if (o.bedType > 6 && (lines[(size_t)r.query].thickStart != 0 || lines[(size_t)r.query].thickEnd != 0)) {
                    o.thickStart = o.start;
                    o.thickEnd = o.end;
                }
                o.write(*out);
            }
        }
        if (!pendingError.empty()) {
            std::string e = pendingError;
            pendingError.clear();
            throw std::runtime_error(e);
        }
    }
}

} // namespace hgx
