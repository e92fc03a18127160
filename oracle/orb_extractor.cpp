// oracle/orb_extractor.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of ORB_SLAM2::ORBextractor, function by function, against reference
// src/ORBextractor.cc (line numbers cited per function). Zero third-party dependencies: the OpenCV
// 2.4 primitives it needs are restated in cvprim.{h,cpp}. PARITY UNPINNED with respect to the
// un-vendored OpenCV arithmetic — the reference holds no test, fixture or golden
// vector for this path; what IS pinned: the constant tables (pattern, umax, quotas, level sizes of
// SURVEY.md §8) and the definitional checks in tests/test_oracle_extractor.py.
//
// Deliberate, documented deviation (SURVEY.md §7 hard part 3): the reference sorts
// (size, ExtractorNode*) pairs, i.e. breaks size ties by heap address (src/ORBextractor.cc:684),
// which is non-deterministic. Here the tie-break is the node creation sequence number (what a
// monotonically growing heap would give).
#include "orb_extractor.h"
#include <list>
#include <cstring>
#include <cassert>

namespace ora {

static const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

static const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

// src/ORBextractor.cc:410-470
OrbExtractor::OrbExtractor(int nf, float sf, int nl, int ini, int mn)
    : nfeatures(nf), nlevels(nl), iniThFAST(ini), minThFAST(mn), scaleFactor(sf) {
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);      // float * double
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mnFeaturesPerLevel.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        mnFeaturesPerLevel[l] = cvRound(nDesired);
        sum += mnFeaturesPerLevel[l];
        nDesired *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    std::memcpy(pattern, kPattern, sizeof(pattern));

    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// src/ORBextractor.cc:1107-1132. The 19-px BORDER_REFLECT_101 frame the reference writes around
// every level is never read on this path (FAST cells start 16 px inside, orientation/descriptor
// patches stay inside the level, the blur runs on a border-less clone, :1085), so levels are kept
// un-padded here.
void OrbExtractor::compute_pyramid(const uint8_t* img, int w, int h, int stride) {
    pyramid.assign(nlevels, Image8());
    for (int level = 0; level < nlevels; ++level) {
        float scale = mvInvScaleFactor[level];
        int sw = cvRound((double)((float)w * scale)), sh = cvRound((double)((float)h * scale));
        if (level == 0) {
            pyramid[0] = Image8(sw, sh);
            for (int y = 0; y < h; y++) std::memcpy(pyramid[0].row(y), img + (size_t)y * stride, w);
        } else {
            resize_linear_8u(pyramid[level - 1], pyramid[level], sw, sh);
        }
    }
}

// src/ORBextractor.cc:77-104
float ic_angle(const Image8& image, float ptx, float pty, const std::vector<int>& u_max) {
    int m_01 = 0, m_10 = 0;
    const int cy = cvRound(pty), cx = cvRound(ptx), step = image.w;
    const uint8_t* center = image.row(cy) + cx;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fastAtan2((float)m_01, (float)m_10);
}

// src/ORBextractor.cc:107-147
void orb_descriptor(const KeyPoint& kpt, const Image8& img, const int8_t* pat, uint8_t* desc) {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = (float)kpt.angle * factorPI;
    // Reference: (float)cos(angle), (float)sin(angle) on a float = libm cosf/sinf. glibc's cosf/sinf
    // are NOT correctly rounded (they differ from the nearest float for 0.04% / 0.1% of the floats in
    // [0, 2pi], measured exhaustively), so the reference's own result depends on its libm build.
    // Deliberate deviation (SURVEY.md §7 hard part 2): a and b are the correctly rounded values,
    // obtained as double cos/sin rounded once to float.
    float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
    const uint8_t* center = img.row(cvRound(kpt.y)) + cvRound(kpt.x);
    const int step = img.w;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int t = 0; t < 8; t++) {
            const int8_t* q = pat + (size_t)(i * 8 + t) * 4;
            // GET_VALUE(idx): center[cvRound(x*b + y*a)*step + cvRound(x*a - y*b)]; int*float
            // products and their sum are separately rounded float operations (no contraction:
            // this file is built with -ffp-contract=off).
            float x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3];
            int t0 = center[cvRound((double)(x0 * b + y0 * a)) * step + cvRound((double)(x0 * a - y0 * b))];
            int t1 = center[cvRound((double)(x1 * b + y1 * a)) * step + cvRound((double)(x1 * a - y1 * b))];
            val |= (t0 < t1) << t;
        }
        desc[i] = (uint8_t)val;
    }
}

namespace {
struct Node {
    std::vector<KeyPoint> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    bool noMore = false;
    int seq = 0;
};
typedef std::list<Node>::iterator NodeIt;

// src/ORBextractor.cc:481-537
void divide(const Node& n, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)std::ceil(static_cast<float>(n.URx - n.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(n.BRy - n.ULy) / 2);
    n1.ULx = n.ULx; n1.ULy = n.ULy; n1.URx = n.ULx + halfX; n1.URy = n.ULy;
    n1.BLx = n.ULx; n1.BLy = n.ULy + halfY; n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = n.URx; n2.URy = n.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = n.BLx; n3.BLy = n.BLy; n3.BRx = n1.BRx; n3.BRy = n.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = n.BRx; n4.BRy = n.BRy;
    for (const KeyPoint& kp : n.keys) {
        if (kp.x < n1.URx) { if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp); }
        else if (kp.y < n1.BRy) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}
} // namespace

// src/ORBextractor.cc:539-763
std::vector<KeyPoint> OrbExtractor::distribute_octree(const std::vector<KeyPoint>& keys, int minX,
                                                      int maxX, int minY, int maxY, int N) const {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    std::vector<KeyPoint> result;
    if (nIni < 1) return result;             // reference would divide by zero; not reachable for w >= h/2
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<NodeIt> bySeq;               // creation sequence -> list position
    int seq = 0;
    std::vector<NodeIt> ini(nIni);
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        nodes.push_back(ni);
        ini[i] = std::prev(nodes.end());
        bySeq.push_back(ini[i]);
    }
    for (const KeyPoint& kp : keys) ini[(size_t)(kp.x / hX)]->keys.push_back(kp);
    for (NodeIt it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->noMore = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }
    bool finish = false;
    std::vector<std::pair<int, int>> sizeAndSeq;     // (size, creation seq) — see header note
    auto push_child = [&](Node& c, int* nToExpand) {
        if (c.keys.empty()) return;
        c.seq = seq++;
        nodes.push_front(c);
        bySeq.push_back(nodes.begin());
        if (c.keys.size() > 1) {
            if (nToExpand) (*nToExpand)++;
            sizeAndSeq.push_back(std::make_pair((int)c.keys.size(), c.seq));
        }
    };
    while (!finish) {
        int prevSize = (int)nodes.size();
        NodeIt it = nodes.begin();
        int nToExpand = 0;
        sizeAndSeq.clear();
        while (it != nodes.end()) {
            if (it->noMore) { ++it; continue; }
            Node n1, n2, n3, n4;
            divide(*it, n1, n2, n3, n4);
            push_child(n1, &nToExpand); push_child(n2, &nToExpand);
            push_child(n3, &nToExpand); push_child(n4, &nToExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if (((int)nodes.size() + nToExpand * 3) > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<std::pair<int, int>> prev = sizeAndSeq;
                sizeAndSeq.clear();
                std::sort(prev.begin(), prev.end());
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    NodeIt nit = bySeq[prev[j].second];
                    Node n1, n2, n3, n4;
                    divide(*nit, n1, n2, n3, n4);
                    push_child(n1, nullptr); push_child(n2, nullptr);
                    push_child(n3, nullptr); push_child(n4, nullptr);
                    nodes.erase(nit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    result.reserve(nodes.size());
    for (const Node& n : nodes) {
        const KeyPoint* best = &n.keys[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if (n.keys[k].response > maxResponse) { best = &n.keys[k]; maxResponse = n.keys[k].response; }
        result.push_back(*best);
    }
    return result;
}

// src/ORBextractor.cc:765-853
void OrbExtractor::compute_keypoints() {
    candidates.assign(nlevels, std::vector<KeyPoint>());
    level_kps.assign(nlevels, std::vector<KeyPoint>());
    const float W = 30;
    for (int level = 0; level < nlevels; ++level) {
        const Image8& im = pyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
        std::vector<KeyPoint>& toDistribute = candidates[level];
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) continue;
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        std::vector<FastKP> cell;
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                fast9_16(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, iniThFAST, cell);
                if (cell.empty()) fast9_16(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, minThFAST, cell);
                for (const FastKP& f : cell) {
                    KeyPoint kp;
                    kp.x = (float)f.x + j * wCell; kp.y = (float)f.y + i * hCell;
                    kp.size = 7.f; kp.angle = -1; kp.response = (float)f.score; kp.octave = 0; kp.class_id = -1;
                    toDistribute.push_back(kp);
                }
            }
        }
        std::vector<KeyPoint>& kps = level_kps[level];
        kps = distribute_octree(toDistribute, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                mnFeaturesPerLevel[level]);
        const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
        for (KeyPoint& kp : kps) {
            kp.x += minBorderX; kp.y += minBorderY; kp.octave = level; kp.size = (float)scaledPatchSize;
        }
    }
    for (int level = 0; level < nlevels; ++level)
        for (KeyPoint& kp : level_kps[level]) kp.angle = ic_angle(pyramid[level], kp.x, kp.y, umax);
}

// src/ORBextractor.cc:1043-1105
int OrbExtractor::extract(const uint8_t* img, int w, int h, int stride,
                          std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc) {
    kps.clear(); desc.clear();
    if (!img || w <= 0 || h <= 0) return 0;          // "if(_image.empty()) return;"
    compute_pyramid(img, w, h, stride);
    compute_keypoints();
    blurred.assign(nlevels, Image8());
    int n = 0;
    for (int l = 0; l < nlevels; l++) n += (int)level_kps[l].size();
    kps.reserve(n); desc.resize((size_t)n * 32);
    int offset = 0;
    for (int level = 0; level < nlevels; ++level) {
        std::vector<KeyPoint> lk = level_kps[level];
        if (lk.empty()) continue;
        gaussian_blur_7x7_s2(pyramid[level], blurred[level]);
        for (size_t i = 0; i < lk.size(); i++)
            orb_descriptor(lk[i], blurred[level], pattern, desc.data() + (size_t)(offset + i) * 32);
        offset += (int)lk.size();
        if (level != 0) {
            float scale = mvScaleFactor[level];
            for (KeyPoint& kp : lk) { kp.x *= scale; kp.y *= scale; }
        }
        kps.insert(kps.end(), lk.begin(), lk.end());
    }
    return n;
}

} // namespace ora
