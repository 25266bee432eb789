// hgxConvert — import an mmap-format HAL file (or re-write an HGX image) as an HGX flat image.
#include "../hgx_image.hpp"
#include <iostream>
int main(int argc, char **argv) {
    if (argc != 3) {
        std::cerr << "usage: hgxConvert <in.hal|in.hgx> <out.hgx>" << std::endl;
        return 1;
    }
    try {
        hgx::Image img = hgx::openAlignmentFile(argv[1]);
        img.validate();
        hgx::writeImage(img, argv[2]);
    } catch (std::exception &e) {
        std::cerr << "hal exception caught: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
