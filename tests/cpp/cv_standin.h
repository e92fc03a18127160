// tests/cpp/cv_standin.h — minimal stand-ins for the few cv:: types the ORBextractor shim header mentions,
// ONLY to compile-test viorb_amd/shim/ORBextractor.h in an image without OpenCV. Not part of the product.
#pragma once
#include <vector>
#include <cstring>
#include <cassert>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
namespace cv {
struct KeyPoint {
    struct Pt { float x, y; } pt; float size, angle, response; int octave, class_id;
    KeyPoint(float x = 0, float y = 0, float s = 0, float a = -1, float r = 0, int o = 0, int c = -1) : size(s), angle(a), response(r), octave(o), class_id(c) { pt.x = x; pt.y = y; }
};
struct Mat {
    int rows = 0, cols = 0; size_t step = 0; unsigned char* data = nullptr; std::vector<unsigned char> buf;
    std::vector<float> fbuf;                                   // CV_32F payload of the few float matrices the tracking shim reads (GetWorldPos)
    template <class T> T& at(int i) { return reinterpret_cast<T*>(fbuf.data())[i]; }
    template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(fbuf.data())[i]; }
    template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(fbuf.data())[(size_t)r * cols + c]; }
    template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(fbuf.data())[(size_t)r * cols + c]; }
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        rows = r; cols = c;
        if (type == CV_32F) { step = (size_t)c * 4; fbuf.assign((size_t)r * c, 0.f); buf.clear(); data = nullptr; }
        else { step = (size_t)c; buf.assign((size_t)r * c, 0); data = buf.data(); }
    }
    bool empty() const { return rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    Mat getMat() const { return *this; }
    void release() { rows = cols = 0; buf.clear(); data = nullptr; }
    Mat rowRange(int a, int b) const { Mat m(b - a, cols, 0); std::memcpy(m.data, data + (size_t)a * step, (size_t)(b - a) * cols); return m; }
    void copyTo(Mat& o) const { o = *this; o.data = o.buf.data(); }
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), step(o.step), buf(o.buf), fbuf(o.fbuf) { data = buf.empty() ? o.data : buf.data(); }
    Mat& operator=(const Mat& o) { rows = o.rows; cols = o.cols; step = o.step; buf = o.buf; fbuf = o.fbuf; data = buf.empty() ? o.data : buf.data(); return *this; }
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
}
