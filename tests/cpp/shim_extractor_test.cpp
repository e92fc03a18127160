// Compile/link test of the drop-in header viorb_amd/shim/ORBextractor.h against libviorb_hip.so.
// Usage: shim_extractor_test [w h]  — with a GPU it extracts a synthetic gradient-noise image and prints the
// keypoint count; without one it checks construction, tables and the empty-image path.
#define VIORB_SHIM_CV_STANDIN
#include "cv_standin.h"
#include "ORBextractor.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    ORB_SLAM2::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    if (ex.GetLevels() != 8 || ex.GetScaleFactors().size() != 8) { std::printf("FAIL tables\n"); return 1; }
    std::vector<cv::KeyPoint> kps; cv::Mat desc, empty, mask;
    ex(empty, mask, kps, desc);                                   // empty image: silent return
    if (!kps.empty()) { std::printf("FAIL empty\n"); return 1; }
    if (viorb_device_count() < 1) { std::printf("OK (no GPU: construction + tables + empty image)\n"); return 0; }
    const int w = argc > 2 ? std::atoi(argv[1]) : 752, h = argc > 2 ? std::atoi(argv[2]) : 480;
    cv::Mat im(h, w, CV_8U);
    unsigned s = 12345;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { s = s * 1664525u + 1013904223u; im.data[(size_t)y * w + x] = (unsigned char)(((x / 16 + y / 16) % 2) * 120 + 60 + (s >> 28)); }
    ex(im, mask, kps, desc);
    if (ex.mvImagePyramid.downloads() != 0) { std::printf("FAIL: pyramid downloaded without being read\n"); return 1; }     // lazy: the mono path never pays for it
    std::printf("OK %zu keypoints, descriptors %dx%d, pyramid[7] %dx%d\n", kps.size(), desc.rows, desc.cols, ex.mvImagePyramid[7].cols, ex.mvImagePyramid[7].rows);
    return kps.size() > 100 && desc.rows == (int)kps.size() ? 0 : 1;
}
