// oracle/cvprim.cpp — TEST INFRASTRUCTURE ONLY. See cvprim.h for scope and the "parity unpinned"
// statement (OpenCV 2.4 is not vendored in the reference).
#include "cvprim.h"
#include <cfloat>
#include <cstring>

namespace ora {

static inline short sat_short_from_float(float v) {
    int i = cvRound((double)v);                      // saturate_cast<short>(float) = cvRound then clamp
    return (short)std::min(std::max(i, -32768), 32767);
}
static inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

void resize_linear_8u(const Image8& src, Image8& dst, int dw, int dh) {
    const int sw = src.w, sh = src.h;
    dst = Image8(dw, dh);
    // cv::resize: inv_scale = dsize/ssize (double), scale = 1./inv_scale.
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int ONE = 2048;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * (size_t)dw), ibeta(2 * (size_t)dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { xmax = std::min(xmax, dx); if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        ialpha[2 * dx]     = sat_short_from_float((1.f - fx) * ONE);
        ialpha[2 * dx + 1] = sat_short_from_float(fx * ONE);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy]     = sat_short_from_float((1.f - fy) * ONE);
        ibeta[2 * dy + 1] = sat_short_from_float(fy * ONE);
    }
    std::vector<int> r0(dw), r1(dw);
    for (int dy = 0; dy < dh; dy++) {
        const int sy0 = clipi(yofs[dy], 0, sh), sy1 = clipi(yofs[dy] + 1, 0, sh);
        const uint8_t* S0 = src.row(sy0);
        const uint8_t* S1 = src.row(sy1);
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (dx < xmax) {
                r0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx + 1] * ialpha[2 * dx + 1];
                r1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx + 1] * ialpha[2 * dx + 1];
            } else {
                r0[dx] = S0[sx] * ONE;
                r1[dx] = S1[sx] * ONE;
            }
        }
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* D = dst.row(dy);
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

void gaussian_kernel_q8(int n, double sigma, int* k) {
    // getGaussianKernel(n, sigma, CV_32F): float taps normalised with a double sum, then
    // convertTo(CV_32S, scale = 256): saturate_cast<int>(tap * 256.f) = cvRound.
    std::vector<float> cf(n);
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cvRound((double)(cf[i] * 256.f));
    }
}

void gaussian_blur_7x7_s2(const Image8& src, Image8& dst) {
    int k[7];
    gaussian_kernel_q8(7, 2.0, k);
    const int w = src.w, h = src.h;
    dst = Image8(w, h);
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src.row(y);
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * S[reflect101(x + i - 3, w)];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = 0; j < 7; j++) s += k[j] * tmp[(size_t)reflect101(y + j - 3, h) * w + x];
            int v = (s + (1 << 15)) >> 16;
            dst.at(y, x) = (uint8_t)std::min(std::max(v, 0), 255);
        }
}

float fastAtan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Bresenham circle of radius 3, as cv::makeOffsets(patternSize = 16).
static const int kCircle[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int fast_corner_strength(const uint8_t* p, int stride) {
    // cv::cornerScore<16> without its pruning `continue`s and without the threshold floor.
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[kCircle[k & 15][0] + kCircle[k & 15][1] * stride];
    int a0 = -1000, b0 = 1000;
    for (int k = 0; k < 16; k++) {
        int a = d[k], b = d[k];
        for (int j = 1; j < 9; j++) { a = std::min(a, d[k + j]); b = std::max(b, d[k + j]); }
        a0 = std::max(a0, a);
        b0 = std::min(b0, b);
    }
    return std::max(a0, -b0) - 1;
}

void fast9_16(const Image8& img, int x0, int y0, int x1, int y1, int threshold,
              std::vector<FastKP>& out) {
    out.clear();
    const int cols = x1 - x0, rows = y1 - y0, stride = img.w;
    if (cols < 7 || rows < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    const int K = 8, N = 25;
    std::vector<uint8_t> score((size_t)cols * rows, 0);
    std::vector<uint8_t> corner((size_t)cols * rows, 0);
    for (int i = 3; i < rows - 3; i++)
        for (int j = 3; j < cols - 3; j++) {
            const uint8_t* p = img.row(y0 + i) + x0 + j;
            const int v = p[0];
            // literal FAST_t<16> segment test: more than K(=8) contiguous circle pixels all
            // darker than v - t, or all brighter than v + t (scan of 25 wrapped positions).
            bool is_corner = false;
            {
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = p[kCircle[k & 15][0] + kCircle[k & 15][1] * stride];
                    if (x < vt) { if (++count > K) { is_corner = true; break; } } else count = 0;
                }
            }
            if (!is_corner) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = p[kCircle[k & 15][0] + kCircle[k & 15][1] * stride];
                    if (x > vt) { if (++count > K) { is_corner = true; break; } } else count = 0;
                }
            }
            if (is_corner) {
                corner[(size_t)i * cols + j] = 1;
                // cornerScore<16>(ptr, pixel, threshold): a0 starts at threshold.
                int s = std::max(fast_corner_strength(p, stride), threshold - 1 + 0);
                // (for a true corner strength >= threshold already; the max() mirrors a0 = threshold)
                score[(size_t)i * cols + j] = (uint8_t)s;
            }
        }
    // 3x3 non-max suppression on the score map (non-corners hold 0), strict '>'.
    for (int i = 3; i < rows - 3; i++)
        for (int j = 3; j < cols - 3; j++) {
            if (!corner[(size_t)i * cols + j]) continue;
            const int s = score[(size_t)i * cols + j];
            const uint8_t* pr = &score[(size_t)(i - 1) * cols + j];
            const uint8_t* cr = &score[(size_t)i * cols + j];
            const uint8_t* nr = &score[(size_t)(i + 1) * cols + j];
            if (s > pr[-1] && s > pr[0] && s > pr[1] && s > cr[-1] && s > cr[1] &&
                s > nr[-1] && s > nr[0] && s > nr[1])
                out.push_back(FastKP{j, i, s});
        }
}

} // namespace ora
