// oracle/cvprim.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the OpenCV 2.4.x primitives the reference's hot path calls. OpenCV is NOT
// vendored under /root/reference (pinned only as "OpenCV 2.4.3+, tested 2.4.11",
// reference CMakeLists.txt:31, README.md:58), so these follow OpenCV 2.4.11's published scalar
// (non-SIMD, non-IPP, non-tegra) algorithms. PARITY UNPINNED against OpenCV itself: no reference
// test or fixture pins any of them; they are pinned here by independent definitional checks in
// tests/test_oracle_*.py.
//
// Reference call sites: src/ORBextractor.cc:81,103,115,119-120,442,456-457,460 (cvRound/cvFloor/
// cvCeil/fastAtan2), :809,814 (FAST), :1086 (GaussianBlur), :1120 (resize), :1122,1127
// (copyMakeBorder — a no-op for this path, see orb_extractor.cpp).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include <algorithm>

namespace ora {

// cvRound: SSE2 cvtsd2si / lrint under the default rounding mode = round-half-to-even.
inline int cvRound(double v) { return (int)std::nearbyint(v); }
// cvFloor / cvCeil as OpenCV 2.4 writes them (via cvRound and a float difference).
inline int cvFloor(double v) { int i = cvRound(v); float d = (float)(v - i); return i - (d < 0); }
inline int cvCeil(double v)  { int i = cvRound(v); float d = (float)(i - v); return i + (d < 0); }

struct Image8 {
    int w = 0, h = 0;
    std::vector<uint8_t> d;          // row-major, stride == w
    Image8() {}
    Image8(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t& at(int y, int x) { return d[(size_t)y * w + x]; }
    uint8_t at(int y, int x) const { return d[(size_t)y * w + x]; }
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
};

// BORDER_REFLECT_101 index (OpenCV borderInterpolate): gfedcb|abcdefgh|gfedcba
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
    return p;
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1, OpenCV 2.4 fixed-point path:
// 11-bit coefficients (INTER_RESIZE_COEF_SCALE = 2048), HResizeLinear -> int rows,
// VResizeLinear: ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
void resize_linear_8u(const Image8& src, Image8& dst, int dw, int dh);

// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1, OpenCV 2.4:
// float kernel from getGaussianKernel -> convertTo(CV_32S, 256) -> separable integer filter ->
// (sum + (1<<15)) >> 16, saturated.
void gaussian_kernel_q8(int ksize, double sigma, int* k);          // integer taps (x256)
void gaussian_blur_7x7_s2(const Image8& src, Image8& dst);

// cv::fastAtan2 (degrees, [0,360)), OpenCV 2.4.9+ scalar form: 7th-order odd polynomial.
float fastAtan2(float y, float x);

// FAST-9-16 corner score exactly as cv::cornerScore<16>: max over the 16 arcs of 9 contiguous
// circle pixels of min(v - p) / min(p - v), minus 1, floored at `threshold`.
struct FastKP { int x, y, score; };
// cv::FAST(img(roi), kps, threshold, nonmaxSuppression=true) on the sub-image
// rows [y0,y1) x cols [x0,x1) of `img`; returns keypoints (coords relative to the sub-image) in
// OpenCV's output order (row-major).
void fast9_16(const Image8& img, int x0, int y0, int x1, int y1, int threshold,
              std::vector<FastKP>& out);
// Direct definitional corner score of one pixel (used by fast9_16): returns max(A,B)-1 where
// A/B are the arc minima above; the pixel is a FAST corner at threshold t iff score >= t.
int fast_corner_strength(const uint8_t* p, int stride);

} // namespace ora
