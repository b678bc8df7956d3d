// image_io_test.cpp -- CPU test helper: load an image with include/xfeat/image_io.h, convert it like the reference's
// Tracking::GrabImage* does (Camera.RGB flag = argv[2]) and write "rows cols channels\n" + the gray bytes to argv[3].
#include <cstdio>
#include <cstdlib>
#include "xfeat/image_io.h"
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    xfeat::Image8 im;
    if (!xfeat::load_image(argv[1], im)) return 1;
    std::vector<unsigned char> g;
    xfeat::to_gray(im, atoi(argv[2]), g);
    FILE* f = fopen(argv[3], "wb");
    fprintf(f, "%d %d %d\n", im.rows, im.cols, im.channels);
    fwrite(g.data(), 1, g.size(), f);
    fclose(f);
    return 0;
}
