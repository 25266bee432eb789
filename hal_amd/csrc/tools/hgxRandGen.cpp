// hgxRandGen — command-line twin of the reference's halRandGen (randgen/halRandGen.cpp:39-56): same
// option names and presets; writes an HGX flat image instead of an HDF5/mmap HAL file.
#include "../hgx_image.hpp"
#include <cstdlib>
#include <cstring>
#include <iostream>

int main(int argc, char **argv) {
    using namespace hgx;
    RandOptions opt;
    randPreset("medium", opt);
    std::string out;
    // two passes so that --preset applies before the per-option overrides, whatever the order
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            auto val = [&]() -> const char * {
                if (i + 1 >= argc) {
                    std::cerr << "missing value for " << a << std::endl;
                    exit(1);
                }
                return argv[++i];
            };
            if (a == "--preset") {
                const char *v = val();
                if (pass == 0 && !randPreset(v, opt)) {
                    std::cerr << " invalid --preset value: " << v << std::endl;
                    return 1;
                }
            } else if (a == "--meanDegree") {
                double v = atof(val());
                if (pass == 1)
                    opt.meanDegree = v;
            } else if (a == "--maxBranchLength") {
                double v = atof(val());
                if (pass == 1)
                    opt.maxBranchLength = v;
            } else if (a == "--minGenomes" || a == "--maxGenomes" || a == "--minSegmentLength" || a == "--maxSegmentLength" ||
                       a == "--minSegments" || a == "--maxSegments") {
                uint64_t v = strtoull(val(), nullptr, 10);
                if (pass == 1) {
                    if (a == "--minGenomes")
                        opt.minGenomes = v;
                    else if (a == "--maxGenomes")
                        opt.maxGenomes = v;
                    else if (a == "--minSegmentLength")
                        opt.minSegmentLength = v;
                    else if (a == "--maxSegmentLength")
                        opt.maxSegmentLength = v;
                    else if (a == "--minSegments")
                        opt.minSegments = v;
                    else
                        opt.maxSegments = v;
                }
            } else if (a == "--seed") {
                int v = atoi(val());
                if (pass == 1)
                    opt.seed = v;
            } else if (a == "--format" || a == "--mmapFileSize") {
                val(); // storage back-end options of the reference: accepted, not applicable
            } else if (a == "--testRand") {
                // parsed but ignored by the reference too (halRandGen.cpp:110)
            } else if (a == "--noDna") {
                opt.withDna = 0;
            } else if (a.rfind("--", 0) == 0) {
                std::cerr << "unknown option " << a << std::endl;
                return 1;
            } else if (pass == 1) {
                out = a;
            }
        }
    }
    if (out.empty()) {
        std::cerr << "usage: hgxRandGen [options] <out.hgx>" << std::endl;
        return 1;
    }
    try {
        Image img = createRandomAlignment(opt);
        img.validate();
        writeImage(img, out);
        std::cerr << img.newick << std::endl;
    } catch (std::exception &e) {
        std::cerr << "Exception caught: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
