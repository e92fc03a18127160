// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE ONLY. extern "C" surface of the CPU oracle so that
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
// Nothing under viorb_amd/ may link, import or call this library.
#include "orb_extractor.h"
#include <cstring>

using namespace ora;

extern "C" {

void* ora_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new OrbExtractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void ora_extractor_destroy(void* h) { delete (OrbExtractor*)h; }

// Returns the number of keypoints (also when it exceeds cap; only min(n,cap) are written).
int ora_extract(void* h, const uint8_t* img, int w, int hgt, int stride, KeyPoint* kps,
                uint8_t* desc, int cap) {
    OrbExtractor* e = (OrbExtractor*)h;
    std::vector<KeyPoint> k; std::vector<uint8_t> d;
    int n = e->extract(img, w, hgt, stride, k, d);
    int m = n < cap ? n : cap;
    if (m > 0) { std::memcpy(kps, k.data(), sizeof(KeyPoint) * m); std::memcpy(desc, d.data(), (size_t)32 * m); }
    return n;
}
void ora_extractor_tables(void* h, float* sf, float* isf, float* s2, float* is2, int* quota, int* umax16) {
    OrbExtractor* e = (OrbExtractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        sf[i] = e->mvScaleFactor[i]; isf[i] = e->mvInvScaleFactor[i];
        s2[i] = e->mvLevelSigma2[i]; is2[i] = e->mvInvLevelSigma2[i];
        quota[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}
int ora_level_size(void* h, int l, int* w, int* hgt) {
    OrbExtractor* e = (OrbExtractor*)h;
    if (l < 0 || l >= (int)e->pyramid.size()) return -1;
    *w = e->pyramid[l].w; *hgt = e->pyramid[l].h; return 0;
}
// which: 0 = pyramid level, 1 = blurred level. dst must hold w*h bytes.
int ora_level_copy(void* h, int l, int which, uint8_t* dst) {
    OrbExtractor* e = (OrbExtractor*)h;
    const std::vector<Image8>& v = which ? e->blurred : e->pyramid;
    if (l < 0 || l >= (int)v.size()) return -1;
    std::memcpy(dst, v[l].d.data(), v[l].d.size());
    return (int)v[l].d.size();
}
// which: 0 = FAST candidates (pre-octree, relative to the 16-px border origin), 1 = level keypoints.
int ora_level_keypoints(void* h, int l, int which, KeyPoint* dst, int cap) {
    OrbExtractor* e = (OrbExtractor*)h;
    const std::vector<std::vector<KeyPoint>>& v = which ? e->level_kps : e->candidates;
    if (l < 0 || l >= (int)v.size()) return -1;
    int n = (int)v[l].size(), m = n < cap ? n : cap;
    if (m > 0) std::memcpy(dst, v[l].data(), sizeof(KeyPoint) * m);
    return n;
}

// --- primitives, exposed one by one for the definitional tests ---------------------------------
void ora_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Image8 s(sw, sh); std::memcpy(s.d.data(), src, (size_t)sw * sh);
    Image8 d; resize_linear_8u(s, d, dw, dh);
    std::memcpy(dst, d.d.data(), (size_t)dw * dh);
}
void ora_gaussian_blur(const uint8_t* src, int w, int h, uint8_t* dst) {
    Image8 s(w, h); std::memcpy(s.d.data(), src, (size_t)w * h);
    Image8 d; gaussian_blur_7x7_s2(s, d);
    std::memcpy(dst, d.d.data(), (size_t)w * h);
}
void ora_gaussian_kernel_q8(int n, double sigma, int* k) { gaussian_kernel_q8(n, sigma, k); }
float ora_fast_atan2(float y, float x) { return fastAtan2(y, x); }
int ora_cv_round(double v) { return cvRound(v); }
// FAST on the sub-image [x0,x1) x [y0,y1); out = (x,y,score) int triples; returns count.
int ora_fast(const uint8_t* img, int w, int h, int x0, int y0, int x1, int y1, int threshold,
             int* out, int cap) {
    Image8 s(w, h); std::memcpy(s.d.data(), img, (size_t)w * h);
    std::vector<FastKP> v; fast9_16(s, x0, y0, x1, y1, threshold, v);
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].score; }
    return n;
}
float ora_ic_angle(const uint8_t* img, int w, int h, float x, float y) {
    Image8 s(w, h); std::memcpy(s.d.data(), img, (size_t)w * h);
    OrbExtractor e(1000, 1.2f, 8, 20, 7);
    return ic_angle(s, x, y, e.umax);
}
void ora_orb_descriptor(const uint8_t* blurred, int w, int h, float x, float y, float angle, uint8_t* desc) {
    Image8 s(w, h); std::memcpy(s.d.data(), blurred, (size_t)w * h);
    OrbExtractor e(1000, 1.2f, 8, 20, 7);
    KeyPoint kp{x, y, 31.f, angle, 0.f, 0, -1};
    orb_descriptor(kp, s, e.pattern, desc);
}
int ora_distribute_octree(const KeyPoint* keys, int n, int minX, int maxX, int minY, int maxY, int N,
                          KeyPoint* out, int cap) {
    OrbExtractor e(1000, 1.2f, 8, 20, 7);
    std::vector<KeyPoint> v(keys, keys + n);
    std::vector<KeyPoint> r = e.distribute_octree(v, minX, maxX, minY, maxY, N);
    int m = (int)r.size();
    for (int i = 0; i < m && i < cap; i++) out[i] = r[i];
    return m;
}

} // extern "C"
