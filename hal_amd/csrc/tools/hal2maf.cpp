// hal2maf — command-line twin of the reference tool (maf/impl/hal2maf.cpp:18-279) for the options the column
// engine implements; columns come from the GPU through libhgx.
#include "../hgx_columns_host.hpp"
#include <vector>
#include <cstring>
#include <fstream>
#include <iostream>

static std::vector<std::string> chop(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t a = 0, b;
    while ((b = s.find(sep, a)) != std::string::npos) {
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    if (a < s.size())
        out.push_back(s.substr(a));
    return out;
}

int main(int argc, char **argv) {
    std::vector<std::string> pos;
    std::string refGenomeName, refSequenceName, rootGenomeName, targetGenomes, refTargetsPath;
    int64_t start = 0, length = 0, maxBlockLen = 1000, maxRefGap = 0, sliceSize = 0;
    std::vector<int> devices; // --gpus n / --devices a,b,...: hal2mafMP.py's slices, dealt to these devices (one process)
    bool noDupes = false, noAncestors = false, onlySequenceNames = false, unique = false, append = false, onlyOrthologs = false,
         keepEmptyRefBlocks = false, global = false, printTree = false;
    int device = 0;
    try {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            auto val = [&]() -> std::string {
                if (i + 1 >= argc)
                    throw std::runtime_error("missing value for " + a);
                return argv[++i];
            };
            if (a == "--refGenome") refGenomeName = val();
            else if (a == "--refSequence") refSequenceName = val();
            else if (a == "--rootGenome") rootGenomeName = val();
            else if (a == "--targetGenomes") targetGenomes = val();
            else if (a == "--start") start = atoll(val().c_str());
            else if (a == "--length") length = atoll(val().c_str());
            else if (a == "--maxBlockLen") maxBlockLen = atoll(val().c_str());
            else if (a == "--maxRefGap") maxRefGap = atoll(val().c_str());
            else if (a == "--device") device = atoi(val().c_str());
            else if (a == "--sliceSize") sliceSize = atoll(val().c_str()); // hal2mafMP.py --sliceSize
            else if (a == "--gpus") {
                const int n = atoi(val().c_str());
                if (n < 1) throw std::runtime_error("--gpus must be at least 1");
                for (int d = 0; d < n; ++d) devices.push_back(d);
            } else if (a == "--devices") {
                const std::string list = val();
                for (size_t at = 0; at < list.size();) {
                    const size_t comma = list.find(',', at);
                    devices.push_back(atoi(list.substr(at, comma == std::string::npos ? std::string::npos : comma - at).c_str()));
                    if (comma == std::string::npos) break;
                    at = comma + 1;
                }
            }
            else if (a == "--noDupes") noDupes = true;
            else if (a == "--noAncestors") noAncestors = true;
            else if (a == "--onlySequenceNames") onlySequenceNames = true;
            else if (a == "--unique") unique = true;
            else if (a == "--append") append = true;
            else if (a == "--onlyOrthologs") onlyOrthologs = true;
            else if (a == "--keepEmptyRefBlocks") keepEmptyRefBlocks = true;
            else if (a == "--refTargets") refTargetsPath = val();
            else if (a == "--global") global = true;
            else if (a == "--printTree") printTree = true;
            else if (a.rfind("--", 0) == 0) throw std::runtime_error("unknown option " + a);
            else pos.push_back(a);
        }
        if (pos.size() != 2)
            throw std::runtime_error("Too few (or many) arguments");
    } catch (std::exception &e) {
        std::cerr << e.what() << "\nusage: hal2maf [options] <halFile> <mafFile|stdout>" << std::endl;
        return 1;
    }
    hgx_alignment *h = nullptr;
    std::vector<hgx_alignment *> clones;
    int rc = 0;
    try {
        char *err = nullptr;
        if (!devices.empty())
            device = devices[0];
        if (hgx_open(pos[0].c_str(), device, &h, &err) != HGX_OK) {
            std::string m = err ? err : "open failed";
            hgx_free(err);
            throw std::runtime_error(m);
        }
        for (size_t d = 1; d < devices.size(); ++d) {
            hgx_alignment *c = nullptr;
            if (hgx_clone_to_device(h, devices[d], &c, &err) != HGX_OK) {
                std::string m = err ? err : "clone failed";
                hgx_free(err);
                throw std::runtime_error(m);
            }
            clones.push_back(c);
        }
        if (hgx_num_genomes(h) == 0)
            throw std::runtime_error("hal alignment is empty");
        std::set<int> targetSet;
        if (!rootGenomeName.empty()) { // hal2maf.cpp:124-131
            int rg = hgx_genome_id(h, rootGenomeName.c_str());
            if (rg < 0)
                throw std::runtime_error("Root genome " + rootGenomeName + ", not found in alignment");
            if (hgx_genome_parent(h, rg) >= 0) {
                std::vector<int> st(1, rg);
                while (!st.empty()) {
                    int g = st.back();
                    st.pop_back();
                    targetSet.insert(g);
                    for (int k = 0; k < hgx_genome_num_children(h, g); ++k)
                        st.push_back(hgx_genome_child(h, g, k));
                }
            }
        }
        for (const std::string &n : chop(targetGenomes, ',')) {
            int g = hgx_genome_id(h, n.c_str());
            if (g < 0)
                throw std::runtime_error("Target genome, " + n + ", not found in alignment");
            targetSet.insert(g);
        }
        int ref = 0;
        if (!refGenomeName.empty()) {
            ref = hgx_genome_id(h, refGenomeName.c_str());
            if (ref < 0)
                throw std::runtime_error("Reference genome, " + refGenomeName + ", not found in alignment");
        } else {
            for (int g = 0; g < hgx_num_genomes(h); ++g)
                if (hgx_genome_parent(h, g) < 0)
                    ref = g;
        }
        if (noAncestors && hgx_genome_num_children(h, ref) != 0 && !global) // hal2maf.cpp:154
            throw std::runtime_error(std::string("Since the reference genome to be used for the MAF is ancestral (") +
                                     hgx_genome_name(h, ref) + "), the --noAncestors option is invalid.  The --refGenome option can be "
                                     "used to specify a different reference.");
        int refSeq = -1;
        if (!refSequenceName.empty()) {
            refSeq = hgx_sequence_lookup(h, ref, refSequenceName.c_str(), nullptr, nullptr);
            if (refSeq < 0)
                throw std::runtime_error("Reference sequence, " + refSequenceName + ", not found in reference genome, " +
                                         hgx_genome_name(h, ref));
        }
        std::vector<char> fileBuffer(4 << 20); // a few large writes instead of one system call per block / line
        std::ofstream mafFile;
        mafFile.rdbuf()->pubsetbuf(fileBuffer.data(), (std::streamsize)fileBuffer.size());
        if (pos[1] != "stdout") {
            mafFile.open(pos[1].c_str(), append ? std::ios::out | std::ios::app : std::ios::out);
            if (!mafFile)
                throw std::runtime_error("Error opening " + pos[1]);
        }
        std::ostream &mafStream = pos[1] != "stdout" ? mafFile : std::cout;
        hgx::MafExport me;
        me.setMaxRefGap(maxRefGap);
        me.setNoDupes(noDupes);
        me.setNoAncestors(noAncestors);
        me.setUcscNames(!onlySequenceNames);
        me.setUnique(unique);
        me.setAppend(append);
        me.setMaxBlockLength(maxBlockLen);
        me.setOnlyOrthologs(onlyOrthologs);
        me.setKeepEmptyRefBlocks(keepEmptyRefBlocks);
        me.setPrintTree(printTree);
        if (!refTargetsPath.empty()) { // hal2mafWithTargets, hal2maf.cpp:105-119
            std::ifstream bedFile;
            if (refTargetsPath != "stdin") {
                bedFile.open(refTargetsPath);
                if (!bedFile)
                    throw std::runtime_error("Error opening " + refTargetsPath);
            }
            me.convertBed(mafStream, h, ref, refTargetsPath != "stdin" ? bedFile : std::cin, targetSet);
        } else if (global) { // hal2maf.cpp:198-199
            me.convertEntireAlignment(mafStream, h);
        } else if (devices.size() > 1 || sliceSize > 0) { // hal2mafMP.py's way: slices of the reference, one export each, put together in order
            std::vector<hgx_alignment *> hs{h};
            hs.insert(hs.end(), clones.begin(), clones.end());
            hgx::MafExportSettings cfg;
            cfg.noDupes = noDupes;
            cfg.noAncestors = noAncestors;
            cfg.ucscNames = !onlySequenceNames;
            cfg.onlyOrthologs = onlyOrthologs;
            cfg.keepEmptyRefBlocks = keepEmptyRefBlocks;
            cfg.unique = unique;
            cfg.maxBlockLength = maxBlockLen;
            cfg.maxRefGap = maxRefGap;
            cfg.printTree = printTree;
            hgx::mafExportSliced(mafStream, hs, ref, refSeq, start, length, sliceSize, cfg, targetSet);
        } else if (refSeq >= 0) {
            me.convertSequence(mafStream, h, ref, refSeq, start, length, targetSet);
        } else {
            for (int s = 0; s < hgx_genome_num_sequences(h, ref); ++s)
                me.convertSequence(mafStream, h, ref, s, start, length, targetSet);
        }
    } catch (std::exception &e) {
        std::cerr << "hal exception caught: " << e.what() << std::endl;
        rc = 1;
    }
    for (hgx_alignment *c : clones)
        hgx_close(c);
    hgx_close(h);
    return rc;
}
