// halAlignmentDepth — command-line twin of the reference tool (alignmentDepth/halAlignmentDepth.cpp:52-213);
// per-column depths come from the GPU through libhgx.
#include "../hgx_columns_host.hpp"
#include <vector>
#include <fstream>
#include <iostream>

int main(int argc, char **argv) {
    std::vector<std::string> pos;
    std::string wigPath = "stdout", refSequenceName, rootGenomeName, targetGenomes;
    int64_t start = 0, length = 0, step = 1;
    bool countDupes = false, noAncestors = false;
    int device = 0;
    std::vector<int> devices; // --gpus n / --devices a,b,...
    try {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            auto val = [&]() -> std::string {
                if (i + 1 >= argc)
                    throw std::runtime_error("missing value for " + a);
                return argv[++i];
            };
            if (a == "--outWiggle") wigPath = val();
            else if (a == "--refSequence") refSequenceName = val();
            else if (a == "--rootGenome") rootGenomeName = val();
            else if (a == "--targetGenomes") targetGenomes = val();
            else if (a == "--start") start = atoll(val().c_str());
            else if (a == "--length") length = atoll(val().c_str());
            else if (a == "--step") step = atoll(val().c_str());
            else if (a == "--device") device = atoi(val().c_str());
            else if (a == "--gpus") { // the scan shared out over the first n devices (one process; hgx_alignment_depth_multi's way)
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
            else if (a == "--countDupes") countDupes = true;
            else if (a == "--noAncestors") noAncestors = true;
            else if (a.rfind("--", 0) == 0) throw std::runtime_error("unknown option " + a);
            else pos.push_back(a);
        }
        if (pos.size() != 2)
            throw std::runtime_error("Too few (or many) arguments");
        if (!rootGenomeName.empty() && !targetGenomes.empty())
            throw std::runtime_error("--rootGenome and --targetGenomes options are  mutually exclusive");
    } catch (std::exception &e) {
        std::cerr << e.what() << "\nusage: halAlignmentDepth [options] <halPath> <refGenome>" << std::endl;
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
        for (size_t d = 1; d < devices.size(); ++d) { // the image once in host memory, its tables on every device
            hgx_alignment *c = nullptr;
            if (hgx_clone_to_device(h, devices[d], &c, &err) != HGX_OK) {
                std::string m = err ? err : "clone failed";
                hgx_free(err);
                throw std::runtime_error(m);
            }
            clones.push_back(c);
        }
        if (hgx_num_genomes(h) == 0)
            throw std::runtime_error("input hal alignmenet is empty");
        std::set<int> targetSet;
        if (!rootGenomeName.empty()) {
            int rg = hgx_genome_id(h, rootGenomeName.c_str());
            if (rg < 0)
                throw std::runtime_error("Root genome, " + rootGenomeName + ", not found in alignment");
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
        size_t a = 0, b;
        while (a < targetGenomes.size()) {
            b = targetGenomes.find(',', a);
            std::string n = targetGenomes.substr(a, b == std::string::npos ? std::string::npos : b - a);
            int g = hgx_genome_id(h, n.c_str());
            if (g < 0)
                throw std::runtime_error("Target genome, " + n + ", not found in alignment");
            targetSet.insert(g);
            if (b == std::string::npos)
                break;
            a = b + 1;
        }
        const int ref = hgx_genome_id(h, pos[1].c_str());
        if (ref < 0)
            throw std::runtime_error("Reference genome, " + pos[1] + ", not found in alignment");
        int refSeq = -1;
        if (!refSequenceName.empty()) {
            refSeq = hgx_sequence_lookup(h, ref, refSequenceName.c_str(), nullptr, nullptr);
            if (refSeq < 0)
                throw std::runtime_error("Reference sequence, " + refSequenceName + ", not found in reference genome, " + pos[1]);
        }
        if (hgx_genome_num_children(h, ref) != 0 && noAncestors)
            throw std::runtime_error("--noAncestors cannot be used when reference genome (" + pos[1] + ") is ancetral");
        std::vector<char> fileBuffer(4 << 20); // a few large writes instead of one system call per block / line
        std::ofstream ofile;
        ofile.rdbuf()->pubsetbuf(fileBuffer.data(), (std::streamsize)fileBuffer.size());
        if (wigPath != "stdout") {
            ofile.open(wigPath.c_str());
            if (!ofile)
                throw std::runtime_error("Error opening output file " + wigPath);
        }
        std::ostream &out = wigPath == "stdout" ? std::cout : ofile;
        hgx::alignmentDepth(out, h, ref, refSeq, targetSet, start, length, step, countDupes, noAncestors, nullptr, clones.empty() ? nullptr : &clones);
    } catch (std::exception &e) {
        std::cerr << "hal exception caught: " << e.what() << std::endl;
        rc = 1;
    }
    for (hgx_alignment *c : clones)
        hgx_close(c);
    hgx_close(h);
    return rc;
}
