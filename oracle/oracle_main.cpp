// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).  Command-line front end used by tests/ and by
// bench.py's cpu_baseline leg.
//   hal_oracle liftover <img.hgx> <srcGenome> <in.bed> <tgtGenome> <out.bed> [--noDupes] [--bedType N] [--coalescenceLimit G] [--stats]
#include "oracle_blockviz.hpp"
#include "oracle_columns.hpp"
#include "oracle_liftover.hpp"
#include <chrono>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

using namespace orc;

static int cmdLiftover(int argc, char **argv) {
    std::vector<std::string> pos;
    bool noDupes = false, stats = false, outPSL = false, outPSLWithName = false;
    int bedType = 0;
    std::string coalName, recordsPath;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--noDupes")
            noDupes = true;
        else if (a == "--records") // (oracle_liftover.hpp: recordsOut)
            recordsPath = argv[++i];
        else if (a == "--coalescenceLimit")
            coalName = argv[++i];
        else if (a == "--stats")
            stats = true;
        else if (a == "--outPSL")
            outPSL = true;
        else if (a == "--outPSLWithName")
            outPSLWithName = true;
        else if (a == "--bedType")
            bedType = atoi(argv[++i]);
        else
            pos.push_back(a);
    }
    if (pos.size() != 5) {
        std::cerr << "usage: hal_oracle liftover <img.hgx> <srcGenome> <in.bed> <tgtGenome> <out.bed>" << std::endl;
        return 1;
    }
    Alignment al = loadImage(pos[0]);
    int src = al.genomeByName(pos[1]), tgt = al.genomeByName(pos[3]);
    if (src < 0 || tgt < 0) {
        std::cerr << "genome not found" << std::endl;
        return 1;
    }
    std::ifstream in(pos[2]);
    std::stringstream inBuf;
    inBuf << in.rdbuf();
    std::ostringstream outBuf;
    Liftover lo;
    int coal = -1;
    if (!coalName.empty()) {
        coal = al.genomeByName(coalName);
        if (coal < 0) {
            std::cerr << "coalescence limit genome not found" << std::endl;
            return 1;
        }
    }
    std::ofstream recordsFile;
    if (!recordsPath.empty()) {
        recordsFile.open(recordsPath, std::ios::binary);
        lo.recordsOut = &recordsFile;
    }
    std::ofstream out(pos[4]);
    try {
        lo.convert(&al, src, &inBuf, tgt, &outBuf, bedType, !noDupes, coal, outPSL, outPSLWithName);
    } catch (...) { // (halLiftover writes as it goes: what was lifted before a malformed line is in the file)
        out << outBuf.str();
        throw;
    }
    out << outBuf.str();
    if (stats)
        std::cout << "{\"intervals\": " << lo.numIntervals << ", \"records\": " << lo.numRecords
                  << ", \"pieces\": " << lo.numMappedPieces << ", \"map_seconds\": " << lo.mapSeconds << "}" << std::endl;
    return 0;
}

static std::vector<std::string> splitCommas(const std::string &s) {
    std::vector<std::string> out;
    size_t a = 0, b;
    while ((b = s.find(',', a)) != std::string::npos) {
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    if (a < s.size())
        out.push_back(s.substr(a));
    return out;
}

// halAlignmentDepth (alignmentDepth/halAlignmentDepth.cpp:52-213) and hal2maf (maf/impl/hal2maf.cpp:18-217) front ends
static int cmdColumns(bool maf, int argc, char **argv) {
    std::vector<std::string> pos;
    std::string refGenome, refSequence, targetGenomes, rootGenome, refTargets;
    i64 start = 0, length = 0, step = 1, maxBlockLen = 1000, maxRefGap = 0;
    bool countDupes = false, noAncestors = false, noDupes = false, onlySequenceNames = false, onlyOrthologs = false, stats = false,
         unique = false, global = false, keepEmptyRefBlocks = false, printTree = false;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--refGenome")
            refGenome = argv[++i];
        else if (a == "--refSequence")
            refSequence = argv[++i];
        else if (a == "--targetGenomes")
            targetGenomes = argv[++i];
        else if (a == "--rootGenome")
            rootGenome = argv[++i];
        else if (a == "--refTargets")
            refTargets = argv[++i];
        else if (a == "--start")
            start = atoll(argv[++i]);
        else if (a == "--length")
            length = atoll(argv[++i]);
        else if (a == "--step")
            step = atoll(argv[++i]);
        else if (a == "--maxBlockLen")
            maxBlockLen = atoll(argv[++i]);
        else if (a == "--maxRefGap")
            maxRefGap = atoll(argv[++i]);
        else if (a == "--countDupes")
            countDupes = true;
        else if (a == "--noAncestors")
            noAncestors = true;
        else if (a == "--noDupes")
            noDupes = true;
        else if (a == "--onlySequenceNames")
            onlySequenceNames = true;
        else if (a == "--onlyOrthologs")
            onlyOrthologs = true;
        else if (a == "--unique")
            unique = true;
        else if (a == "--keepEmptyRefBlocks")
            keepEmptyRefBlocks = true;
        else if (a == "--printTree")
            printTree = true;
        else if (a == "--global")
            global = true;
        else if (a == "--stats")
            stats = true;
        else
            pos.push_back(a);
    }
    // depth: <img> <refGenome> <out.wig>;  maf: <img> <out.maf>
    if ((maf && pos.size() != 2) || (!maf && pos.size() != 3)) {
        std::cerr << "usage: hal_oracle depth <img> <refGenome> <out.wig> [opts] | hal_oracle maf <img> <out.maf> [opts]" << std::endl;
        return 1;
    }
    Alignment al = loadImage(pos[0]);
    if (!maf)
        refGenome = pos[1];
    int ref = refGenome.empty() ? al.root() : al.genomeByName(refGenome);
    if (ref < 0)
        throw std::runtime_error("Reference genome, " + refGenome + ", not found in alignment");
    std::set<int> targetSet;
    if (!rootGenome.empty()) {
        int rg = al.genomeByName(rootGenome);
        if (rg < 0)
            throw std::runtime_error("Root genome " + rootGenome + ", not found in alignment");
        if (rg != al.root()) { // getGenomesInSubTree, api/impl/halCommon.cpp:189-195
            std::vector<int> st(1, rg);
            while (!st.empty()) {
                int g = st.back();
                st.pop_back();
                targetSet.insert(g);
                for (int c : al.genomes[(size_t)g].children)
                    st.push_back(c);
            }
        }
    }
    for (const std::string &n : splitCommas(targetGenomes)) {
        int g = al.genomeByName(n);
        if (g < 0)
            throw std::runtime_error("Target genome, " + n + ", not found in alignment");
        targetSet.insert(g);
    }
    int seq = -1;
    if (!refSequence.empty()) {
        const Sequence *s = al.genomes[(size_t)ref].seqByName(refSequence);
        if (!s)
            throw std::runtime_error("Reference sequence, " + refSequence + ", not found in reference genome");
        seq = (int)(s - al.genomes[(size_t)ref].seqs.data());
    }
    if (noAncestors && !al.genomes[(size_t)ref].children.empty() && !global) // hal2maf.cpp:154
        throw std::runtime_error("--noAncestors cannot be used when the reference genome is ancestral");
    std::ofstream out(maf ? pos[1] : pos[2]);
    std::ostringstream buf;
    double seconds = 0;
    size_t columns = 0;
    if (!maf) {
        auto t0 = std::chrono::steady_clock::now();
        printDepthGenome(buf, al, ref, seq, targetSet, start, length, step, countDupes, noAncestors);
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (char c : buf.str())
            columns += c == '\n';
    } else {
        MafExport me;
        me.noDupes = noDupes;
        me.noAncestors = noAncestors;
        me.ucscNames = !onlySequenceNames;
        me.onlyOrthologs = onlyOrthologs;
        me.keepEmptyRefBlocks = keepEmptyRefBlocks;
        me.printTree = printTree;
        me.unique = unique;
        me.maxRefGap = maxRefGap;
        me.maxBlockLength = maxBlockLen <= 0 ? std::numeric_limits<i64>::max() : maxBlockLen;
        if (global) { // hal2maf.cpp:198-199
            me.convertEntireAlignment(buf, al);
        } else if (!refTargets.empty()) {
            // MafBed::visitLine (maf/impl/halMafBed.cpp:24-52) over BedScanner::scan
            std::ifstream bedIn(refTargets);
            BedLine bedLine;
            std::string lineBuffer;
            size_t lineNumber = 0;
            auto skipWs = [&]() {
                while (bedIn.good() && std::isspace((char)bedIn.peek()))
                    bedIn.get();
            };
            skipWs();
            while (bedIn.good()) {
                ++lineNumber;
                bedLine.read(bedIn, lineBuffer, 0);
                const Sequence *rs = al.genomes[(size_t)ref].seqByName(bedLine._chrName);
                if (rs == nullptr) {
                    std::cerr << "Line " << lineNumber << ": BED sequence " << bedLine._chrName << " not found" << std::endl;
                } else {
                    const int si = (int)(rs - al.genomes[(size_t)ref].seqs.data());
                    if (bedLine._bedType <= 9) {
                        if (bedLine._end <= bedLine._start || bedLine._end > rs->length)
                            std::cerr << "Line " << lineNumber << ": BED coordinates invalid\n";
                        else
                            me.convertSequence(buf, al, ref, si, bedLine._start, bedLine._end - bedLine._start, targetSet);
                    } else {
                        for (size_t k = 0; k < bedLine._blocks.size(); ++k) {
                            const BedBlock &b = bedLine._blocks[k];
                            if (b._length == 0 || bedLine._start + b._start + b._length >= rs->length)
                                std::cerr << "Line " << lineNumber << ", block " << k << ": BED coordinates invalid\n";
                            else
                                me.convertSequence(buf, al, ref, si, bedLine._start + b._start, b._length, targetSet);
                        }
                    }
                }
                skipWs();
            }
        } else if (seq >= 0) {
            me.convertSequence(buf, al, ref, seq, start, length, targetSet);
        } else {
            for (size_t s = 0; s < al.genomes[(size_t)ref].seqs.size(); ++s)
                me.convertSequence(buf, al, ref, (int)s, start, length, targetSet);
        }
        seconds = me.seconds;
        columns = me.numColumns;
    }
    out << buf.str();
    if (stats)
        std::cout << "{\"columns\": " << columns << ", \"seconds\": " << seconds << "}" << std::endl;
    return 0;
}

// hal_oracle blocks <img.hgx> <refGenome> <queryGenome> <absFirst> <absLast> [--reversed] [--noDupes] [--minLength N]
//                   [--coalescenceLimit G]
// prints the members of BlockMapper::getMap() in set order: target sequence, target range (sequence relative, forward,
// end exclusive), forward source start (genome coordinate), source strand, target strand
static int cmdBlocks(int argc, char **argv) {
    std::vector<std::string> pos;
    bool reversed = false, noDupes = false;
    i64 minLength = 0;
    std::string coalName;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--reversed")
            reversed = true;
        else if (a == "--noDupes")
            noDupes = true;
        else if (a == "--minLength")
            minLength = atoll(argv[++i]);
        else if (a == "--coalescenceLimit")
            coalName = argv[++i];
        else
            pos.push_back(a);
    }
    if (pos.size() != 5) {
        std::cerr << "usage: hal_oracle blocks <img.hgx> <refGenome> <queryGenome> <absFirst> <absLast>" << std::endl;
        return 1;
    }
    Alignment al = loadImage(pos[0]);
    int ref = al.genomeByName(pos[1]), query = al.genomeByName(pos[2]);
    int coal = coalName.empty() ? -1 : al.genomeByName(coalName);
    if (ref < 0 || query < 0 || (!coalName.empty() && coal < 0)) {
        std::cerr << "genome not found" << std::endl;
        return 1;
    }
    MSegSet segs;
    blockMap(al, ref, query, atoll(pos[3].c_str()), atoll(pos[4].c_str()), reversed, !noDupes, minLength, coal, segs);
    for (MSegSet::iterator i = segs.begin(); i != segs.end(); ++i) {
        const SegIt &t = (*i)->tgt, &s = (*i)->src;
        const Sequence *seq = t.getSequence();
        const i64 tLo = std::min(t.getStartPosition(), t.getEndPosition()), tHi = std::max(t.getStartPosition(), t.getEndPosition());
        const i64 sLo = std::min(s.getStartPosition(), s.getEndPosition());
        std::cout << seq->name << '\t' << tLo - seq->start << '\t' << tHi + 1 - seq->start << '\t' << sLo << '\t' << (s.rev ? '-' : '+')
                  << '\t' << (t.rev ? '-' : '+') << '\n';
    }
    return 0;
}

// hal_oracle blockviz <img.hgx> <qSpecies> <tSpecies> <tChrom> <tStart> <tEnd> [--doSeq] [--dupMode 0|1|2] [--noAdj] [--tReversed]
//                     [--coalescenceLimit G]
// = blockVizTest --verbose (blockViz/tests/blockVizTest.cpp:200-236: dupMode HAL_QUERY_AND_TARGET_DUPS and mapBackAdjacencies 1
// unless told otherwise): the blocks, then the target dupe lists, in that program's print format
static int cmdBlockViz(int argc, char **argv) {
    std::vector<std::string> pos;
    bool doSeq = false, adj = true, tReversed = false;
    int dupMode = VIZ_QUERY_AND_TARGET_DUPS;
    std::string coalName, rangesPath, outPath;
    bool stats = false;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--ranges")
            rangesPath = argv[++i];
        else if (a == "--out")
            outPath = argv[++i];
        else if (a == "--stats")
            stats = true;
        else if (a == "--doSeq")
            doSeq = true;
        else if (a == "--noAdj")
            adj = false;
        else if (a == "--tReversed")
            tReversed = true;
        else if (a == "--dupMode")
            dupMode = atoi(argv[++i]);
        else if (a == "--coalescenceLimit")
            coalName = argv[++i];
        else
            pos.push_back(a);
    }
    if (!rangesPath.empty() && pos.size() == 4) {
        // hal_oracle blockviz <img> <qSpecies> <tSpecies> <tChrom> --ranges <file of "tStart tEnd" lines> --out <file> [--stats]:
        // one call per line (bench.py's CPU baseline beside features.blocks_in_target_range); every result behind a "# tStart tEnd" line
        Alignment al = loadImage(pos[0]);
        const int q = al.genomeByName(pos[1]), t = al.genomeByName(pos[2]);
        const int coal = coalName.empty() ? -1 : al.genomeByName(coalName);
        if (q < 0 || t < 0 || (!coalName.empty() && coal < 0)) {
            std::cerr << "genome not found" << std::endl;
            return 1;
        }
        std::ifstream in(rangesPath);
        std::vector<std::pair<long long, long long>> ranges;
        for (long long a, b; in >> a >> b;)
            ranges.emplace_back(a, b);
        std::ofstream out(outPath);
        double seconds = 0;
        for (auto &r : ranges) {
            auto t0 = std::chrono::steady_clock::now();
            VizResults res = getBlocksInTargetRange(al, q, t, pos[3], r.first, r.second, tReversed, doSeq, dupMode, adj, coal);
            seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            out << "# " << r.first << " " << r.second << "\n";
            printVizResults(out, res, doSeq);
        }
        if (stats)
            std::cout << "{\"ranges\": " << ranges.size() << ", \"seconds\": " << seconds << "}" << std::endl;
        return 0;
    }
    if (pos.size() != 6) {
        std::cerr << "usage: hal_oracle blockviz <img.hgx> <qSpecies> <tSpecies> <tChrom> <tStart> <tEnd>" << std::endl;
        return 1;
    }
    Alignment al = loadImage(pos[0]);
    const int q = al.genomeByName(pos[1]), t = al.genomeByName(pos[2]);
    const int coal = coalName.empty() ? -1 : al.genomeByName(coalName);
    if (q < 0 || t < 0 || (!coalName.empty() && coal < 0)) {
        std::cerr << "genome not found" << std::endl;
        return 1;
    }
    VizResults r = getBlocksInTargetRange(al, q, t, pos[3], atoll(pos[4].c_str()), atoll(pos[5].c_str()), tReversed, doSeq, dupMode, adj, coal);
    printVizResults(std::cout, r, doSeq);
    return 0;
}

// hal_oracle columns <img.hgx> <refGenome>: every column of the reference genome's first sequence, one line per
// column: "<col>" then " <genome>:<position>:<+|->" per base in ColumnMap order (sequences in SequenceLess order,
// bases of a sequence in insertion order); what a loop over getColumnIterator() / toRight() / getColumnMap() sees.
static int cmdColumnRows(int argc, char **argv) {
    std::vector<std::string> pos;
    i64 maxInsertLength = 0;
    bool noDupes = false, noAncestors = false, unique = false;
    std::string batches, targetGenomes;
    i64 chunk = 1 << 21;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--maxRefGap")
            maxInsertLength = atoll(argv[++i]);
        else if (a == "--batches") // the columns as a recording of device batches instead of text (below)
            batches = argv[++i];
        else if (a == "--chunk")
            chunk = atoll(argv[++i]);
        else if (a == "--targetGenomes")
            targetGenomes = argv[++i];
        else if (a == "--noDupes")
            noDupes = true;
        else if (a == "--noAncestors")
            noAncestors = true;
        else if (a == "--unique")
            unique = true;
        else
            pos.push_back(a);
    }
    if (pos.size() != 2) {
        std::cerr << "usage: hal_oracle columns <img.hgx> <refGenome> [--maxRefGap N] [--noDupes] [--noAncestors] [--unique]" << std::endl;
        return 1;
    }
    Alignment al = loadImage(pos[0]);
    const int ref = al.genomeByName(pos[1]);
    if (ref < 0) {
        std::cerr << "genome not found" << std::endl;
        return 1;
    }
    if (!batches.empty()) {
        // A profiling aid for hal2maf's host side on a machine without a GPU: the plain export's columns (no --unique, no
        // --maxRefGap) in the layout the library's profiling build records its device batches in (hal_amd/csrc/hgx_columns_host.cpp:
        // HGX_MAF_DUMP / HGX_MAF_REPLAY; make hostprof-lib) — per chunk of columns {first column, columns, heads + 1, rows}, a byte
        // per column (1: the column does not continue the one before base by base), the heads' row offsets, the heads' rows
        // {position, genome, reversed} in the column map's order.
        if (maxInsertLength != 0)
            throw std::runtime_error("--batches: not with --maxRefGap (those columns go one by one through the iterator's replay)");
        struct Row {
            i64 pos;
            int32_t genome;
            uint8_t rev;
            char base;
            uint8_t pad[2];
        };
        static_assert(sizeof(Row) == 16, "ColumnRowHost");
        std::ofstream out(batches, std::ios::binary);
        std::set<int> targetSet; // hal2maf --targetGenomes (the reference is added by the iterator itself)
        for (size_t a0 = 0; a0 < targetGenomes.size();) {
            size_t b0 = targetGenomes.find(',', a0);
            if (b0 == std::string::npos)
                b0 = targetGenomes.size();
            const int g = al.genomeByName(targetGenomes.substr(a0, b0 - a0));
            if (g < 0)
                throw std::runtime_error("--targetGenomes: genome not found");
            targetSet.insert(g);
            a0 = b0 + 1;
        }
        i64 total = 0;
        for (const Sequence &Sq : al.genomes[(size_t)ref].seqs) { // (an export of its own per sequence, as hal2maf makes them)
            if (Sq.length == 0)
                continue;
            // per column of the sequence: 0 continues the column before base by base, 1 a head, and with --unique 2 a column the
            // iterator passes over (nextFreeIndex: its reference base was in an earlier column), 3 a column that is walked but not
            // written (not canonical on the reference); heads and 3s have their rows
            std::vector<uint8_t> head((size_t)Sq.length, unique ? 2 : 1);
            std::vector<std::vector<Row>> rowsOf((size_t)Sq.length);
            ColumnIterator col(&al, ref, targetSet.empty() ? nullptr : &targetSet, Sq.start, Sq.start + Sq.length - 1, noDupes, noAncestors, false, unique, 0);
            i64 prevWritten = -2;
            std::vector<Row> prev, cur;
            for (;;) {
                cur.clear();
                for (auto &kv : col.colMap)
                    for (const Dna &d : kv.second)
                        cur.push_back(Row{d.pos, (int32_t)d.g, (uint8_t)(d.rev ? 1 : 0), 'N', {0, 0}});
                const i64 c = col.refSequencePosition(); // (sequence relative)
                if (c < 0 || c >= Sq.length)
                    throw std::runtime_error("--batches: a column outside the sequence");
                if (unique && !col.isCanonicalOnRef()) {
                    head[(size_t)c] = 3;
                    rowsOf[(size_t)c] = cur;
                } else {
                    bool continues = prevWritten == c - 1 && cur.size() == prev.size();
                    for (size_t i = 0; continues && i < cur.size(); ++i)
                        continues = cur[i].genome == prev[i].genome && cur[i].rev == prev[i].rev && cur[i].pos == prev[i].pos + (cur[i].rev ? -1 : 1);
                    head[(size_t)c] = continues ? 0 : 1;
                    if (!continues)
                        rowsOf[(size_t)c] = cur;
                    prevWritten = c;
                    prev.swap(cur);
                }
                if (col.lastColumn())
                    break;
                col.toRight();
            }
            for (i64 done = 0; done < Sq.length; done += chunk) { // the batches: the first written column of each is a head
                const i64 n = std::min(chunk, Sq.length - done);
                std::vector<uint32_t> headOff(1, 0);
                std::vector<Row> rows;
                std::vector<uint8_t> marks(head.begin() + done, head.begin() + done + n);
                for (i64 i = 0; i < n; ++i) {
                    if (marks[(size_t)i] == 0 && (i == 0 || (marks[(size_t)i - 1] != 0 && marks[(size_t)i - 1] != 1))) {
                        // (a continuation at a batch's beginning: the device sees no column before it; its rows are the head's, advanced)
                        marks[(size_t)i] = 1;
                        i64 h = done + i;
                        while (head[(size_t)h] == 0)
                            --h;
                        std::vector<Row> r = rowsOf[(size_t)h];
                        for (Row &x : r)
                            x.pos += (x.rev ? -1 : 1) * (done + i - h);
                        rows.insert(rows.end(), r.begin(), r.end());
                        headOff.push_back((uint32_t)rows.size());
                        continue;
                    }
                    if (marks[(size_t)i] & 1) {
                        const std::vector<Row> &r = rowsOf[(size_t)(done + i)];
                        rows.insert(rows.end(), r.begin(), r.end());
                        headOff.push_back((uint32_t)rows.size());
                    }
                }
                const uint64_t hd[4] = {(uint64_t)done, (uint64_t)n, headOff.size(), rows.size()};
                out.write((const char *)hd, 32);
                out.write((const char *)marks.data(), (std::streamsize)marks.size());
                out.write((const char *)headOff.data(), (std::streamsize)(4 * headOff.size()));
                out.write((const char *)rows.data(), (std::streamsize)(16 * rows.size()));
            }
            total += Sq.length;
        }
        std::cerr << "columns " << total << std::endl;
        return 0;
    }
    const Sequence &S = al.genomes[(size_t)ref].seqs[0];
    ColumnIterator col(&al, ref, nullptr, S.start, S.start + S.length - 1, noDupes, noAncestors, false, unique, maxInsertLength);
    // (with a stack — --maxRefGap — or a visit cache the number of columns is not the sequence's length: until lastColumn())
    for (i64 c = 0;; ++c) {
        std::cout << c;
        for (auto &kv : col.colMap)
            for (const Dna &d : kv.second)
                std::cout << ' ' << al.genomes[(size_t)d.g].name << ':' << d.pos << ':' << (d.rev ? '-' : '+');
        std::cout << '\n';
        if (col.lastColumn())
            break;
        col.toRight();
    }
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc >= 2 && std::string(argv[1]) == "columns")
            return cmdColumnRows(argc - 2, argv + 2);
        if (argc >= 2 && std::string(argv[1]) == "blocks")
            return cmdBlocks(argc - 2, argv + 2);
        if (argc >= 2 && std::string(argv[1]) == "blockviz")
            return cmdBlockViz(argc - 2, argv + 2);
        if (argc >= 2 && std::string(argv[1]) == "liftover")
            return cmdLiftover(argc - 2, argv + 2);
        if (argc >= 2 && std::string(argv[1]) == "depth")
            return cmdColumns(false, argc - 2, argv + 2);
        if (argc >= 2 && std::string(argv[1]) == "maf")
            return cmdColumns(true, argc - 2, argv + 2);
        std::cerr << "usage: hal_oracle liftover ..." << std::endl;
        return 1;
    } catch (std::exception &e) {
        std::cerr << "hal exception caught: " << e.what() << std::endl;
        return 1;
    }
}
