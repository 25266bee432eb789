// ORACLE — TEST INFRASTRUCTURE ONLY (see hal_oracle.hpp).  Command-line front end used by tests/ and by
// bench.py's cpu_baseline leg.
//   hal_oracle liftover <img.hgx> <srcGenome> <in.bed> <tgtGenome> <out.bed> [--noDupes] [--bedType N] [--stats]
#include "oracle_liftover.hpp"
#include <fstream>
#include <iostream>
#include <sstream>

using namespace orc;

static int cmdLiftover(int argc, char **argv) {
    std::vector<std::string> pos;
    bool noDupes = false, stats = false;
    int bedType = 0;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--noDupes")
            noDupes = true;
        else if (a == "--stats")
            stats = true;
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
    lo.convert(&al, src, &inBuf, tgt, &outBuf, bedType, !noDupes);
    std::ofstream out(pos[4]);
    out << outBuf.str();
    if (stats)
        std::cout << "{\"intervals\": " << lo.numIntervals << ", \"records\": " << lo.numRecords
                  << ", \"pieces\": " << lo.numMappedPieces << ", \"map_seconds\": " << lo.mapSeconds << "}" << std::endl;
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc >= 2 && std::string(argv[1]) == "liftover")
            return cmdLiftover(argc - 2, argv + 2);
        std::cerr << "usage: hal_oracle liftover ..." << std::endl;
        return 1;
    } catch (std::exception &e) {
        std::cerr << "hal exception caught: " << e.what() << std::endl;
        return 1;
    }
}
