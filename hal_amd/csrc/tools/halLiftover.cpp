// halLiftover — command-line twin of the reference tool (liftover/impl/halLiftoverMain.cpp:17-152):
// same arguments, options and messages; the mapping runs on the GPU through libhgx.
#include "../hgx_liftover_host.hpp"
#include <vector>
#include <cstring>
#include <fstream>
#include <iostream>

static void usage() {
    std::cerr << "halLiftover [--noDupes] [--append] [--coalescenceLimit <genome>] [--outPSL] [--outPSLWithName]\n"
                 "            [--bedType <3..12>] [--device <ordinal> | --gpus <n> | --devices <a,b,...>]\n"
                 "            <halFile> <srcGenome> <srcBed> <tgtGenome> <tgtBed>\n"
                 "Map BED or PSL genome interval coordinates between two genomes.\n";
}

int main(int argc, char **argv) {
    std::vector<std::string> pos;
    std::string coalescenceLimitName;
    bool noDupes = false, append = false, outPSL = false, outPSLWithName = false;
    int bedType = 0, device = 0;
    std::vector<int> devices; // --gpus n / --devices a,b,...: the input's lines are shared out over these (one process)
    try {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            auto val = [&]() -> std::string {
                if (i + 1 >= argc)
                    throw std::runtime_error("missing value for " + a);
                return argv[++i];
            };
            if (a == "--noDupes")
                noDupes = true;
            else if (a == "--append")
                append = true;
            else if (a == "--outPSL")
                outPSL = true;
            else if (a == "--outPSLWithName")
                outPSLWithName = true;
            else if (a == "--coalescenceLimit")
                coalescenceLimitName = val();
            else if (a == "--bedType") {
                bedType = atoi(val().c_str());
                if (bedType < 3 || bedType > 12)
                    throw std::runtime_error("--bedType must be between 3 and 12");
            } else if (a == "--device")
                device = atoi(val().c_str());
            else if (a == "--gpus") {
                const int n = atoi(val().c_str());
                if (n < 1)
                    throw std::runtime_error("--gpus must be at least 1");
                for (int d = 0; d < n; ++d)
                    devices.push_back(d);
            } else if (a == "--devices") {
                const std::string list = val();
                for (size_t at = 0; at < list.size();) {
                    const size_t comma = list.find(',', at);
                    devices.push_back(atoi(list.substr(at, comma == std::string::npos ? std::string::npos : comma - at).c_str()));
                    if (comma == std::string::npos)
                        break;
                    at = comma + 1;
                }
            }
            else if (a == "--format" || a == "--mmapFileSize" || a == "--hdf5InMemory" || a == "--cacheBytes" || a == "--cacheMDC" ||
                     a == "--cacheRDC" || a == "--cacheW0") {
                if (a != "--hdf5InMemory")
                    val(); // back-end options of the reference: accepted, not applicable
            } else if (a.rfind("--", 0) == 0)
                throw std::runtime_error("unknown option " + a);
            else
                pos.push_back(a);
        }
        if (pos.size() != 5)
            throw std::runtime_error("Too few (or many) arguments");
    } catch (std::exception &e) {
        std::cerr << e.what() << std::endl;
        usage();
        return 1;
    }
    hgx_alignment *h = nullptr;
    std::vector<hgx_alignment *> clones;
    int rc = 0;
    try {
        if (outPSLWithName)
            outPSL = true;
        char *err = nullptr;
        if (!devices.empty())
            device = devices[0];
        if (hgx_open(pos[0].c_str(), device, &h, &err) != HGX_OK) {
            std::string m = err ? err : "open failed";
            hgx_free(err);
            throw std::runtime_error(m);
        }
        if (hgx_num_genomes(h) == 0)
            throw std::runtime_error("hal alignment is empty");
        for (size_t d = 1; d < devices.size(); ++d) { // the image once in host memory, its tables on every device
            hgx_alignment *c = nullptr;
            if (hgx_clone_to_device(h, devices[d], &c, &err) != HGX_OK) {
                std::string m = err ? err : "device clone failed";
                hgx_free(err);
                throw std::runtime_error(m);
            }
            clones.push_back(c);
        }
        const int src = hgx_genome_id(h, pos[1].c_str());
        if (src < 0)
            throw std::runtime_error("srcGenome, " + pos[1] + ", not found in alignment");
        const int tgt = hgx_genome_id(h, pos[3].c_str());
        if (tgt < 0)
            throw std::runtime_error("tgtGenome, " + pos[3] + ", not found in alignment");
        int coal = -1;
        if (!coalescenceLimitName.empty()) {
            coal = hgx_genome_id(h, coalescenceLimitName.c_str());
            if (coal < 0)
                throw std::runtime_error("coalescence limit genome " + coalescenceLimitName + " not found in alignment\n");
        }
        std::ifstream srcBed;
        std::istream *srcBedPtr = &std::cin;
        if (pos[2] != "stdin") {
            srcBed.open(pos[2].c_str());
            srcBedPtr = &srcBed;
            if (!srcBed)
                throw std::runtime_error("Error opening srcBed, " + pos[2]);
        }
        std::vector<char> fileBuffer(4 << 20); // a few large writes instead of one system call per block / line
        std::ofstream tgtBed;
        tgtBed.rdbuf()->pubsetbuf(fileBuffer.data(), (std::streamsize)fileBuffer.size());
        std::ostream *tgtBedPtr = &std::cout;
        if (pos[4] != "stdout") {
            tgtBed.open(pos[4].c_str(), append ? std::ios::out | std::ios::app : std::ios::out);
            tgtBedPtr = &tgtBed;
            if (!tgtBed)
                throw std::runtime_error("Error opening tgtBed, " + pos[4]);
        }
        hgx::Liftover liftover;
        liftover.moreDevices = clones;
        liftover.convert(h, src, srcBedPtr, tgt, tgtBedPtr, bedType, !noDupes, outPSL, outPSLWithName, coal);
    } catch (std::exception &e) {
        std::cerr << "hal exception caught: " << e.what() << std::endl;
        rc = 1;
    }
    for (hgx_alignment *c : clones)
        hgx_close(c);
    hgx_close(h);
    return rc;
}
