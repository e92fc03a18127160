// viorb_amd/csrc/octree_arrays.h — ORBextractor::DistributeOctTree (reference
// src/ORBextractor.cc:539-763, DivideNode :481-537) re-formulated on flat arrays so that the same
// steps run as one wavefront per (image, level) on the device (k_octree in orb_extractor.hip).
//
// Formulation (shared by the host function below and the kernel):
//   * keys are packed u32 (x:12 | y:12<<12 | score:8<<24), coordinates relative to the 16-px border
//     origin; every node owns a contiguous segment [begin, begin+count) of a permutation array,
//     in the reference's push order; DivideNode = stable 4-way partition of that segment into the
//     other of two ping-pong permutation buffers.
//   * the reference's std::list with push_front becomes an array that only grows at its END:
//     array index ascending == list back-to-front, so "iterate the list from begin()" == walk the
//     array downwards, and erase == tombstone + stable compaction between rounds.
//   * the largest-first phase sorts (count<<16 | array_index): the index tie-break is the node
//     creation order (the reference compares heap pointers there, src/ORBextractor.cc:684 — the
//     deterministic replacement documented in DESIGN.md).
// This host version exists only behind the test hook viorb_debug_octree_host(): the CPU parity tests
// validate the formulation against the list-based oracle without a GPU. The product never runs it.
#pragma once
#include <stdint.h>
#include <vector>
#include <algorithm>
#include <math.h>

namespace viorb {

struct __attribute__((aligned(16))) OctNode {
    int16_t x0, x1, y0, y1;
    uint16_t begin, count;
    uint16_t flags;            // bit0 = permutation buffer, bit1 = noMore, bit2 = dead
    uint16_t pad;
};
enum { OCT_BUF = 1, OCT_NOMORE = 2, OCT_DEAD = 4 };

static inline uint32_t oct_pack(int x, int y, int score) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)score << 24); }
static inline int oct_x(uint32_t k) { return (int)(k & 0xfff); }
static inline int oct_y(uint32_t k) { return (int)((k >> 12) & 0xfff); }
static inline int oct_s(uint32_t k) { return (int)(k >> 24); }

// keys: candidates in push order. width/height = maxX-minX / maxY-minY. Returns the kept keys in the
// reference's output order (list order of the final nodes).
inline std::vector<uint32_t> distribute_octree_arrays(const std::vector<uint32_t>& keys, int width,
                                                      int height, int N) {
    std::vector<uint32_t> out;
    const int n = (int)keys.size();
    const int nIni = (int)roundf((float)width / (float)height);
    if (n == 0 || nIni < 1) return out;
    const float hX = (float)width / (float)nIni;
    std::vector<uint16_t> perm[2];
    perm[0].resize(n); perm[1].resize(n);
    std::vector<OctNode> nd;
    // roots: counting sort of the keys by root index (stable)
    {
        std::vector<int> cnt(nIni + 1, 0);
        for (int i = 0; i < n; i++) cnt[(int)((float)oct_x(keys[i]) / hX) + 1]++;
        for (int r = 0; r < nIni; r++) cnt[r + 1] += cnt[r];
        std::vector<int> pos(cnt.begin(), cnt.end() - 1);
        for (int i = 0; i < n; i++) perm[0][pos[(int)((float)oct_x(keys[i]) / hX)]++] = (uint16_t)i;
        for (int r = 0; r < nIni; r++) {
            int c = cnt[r + 1] - cnt[r];
            if (c == 0) continue;
            OctNode o;
            o.x0 = (int16_t)(int)(hX * (float)r); o.x1 = (int16_t)(int)(hX * (float)(r + 1));
            o.y0 = 0; o.y1 = (int16_t)height;
            o.begin = (uint16_t)cnt[r]; o.count = (uint16_t)c; o.flags = (c == 1) ? OCT_NOMORE : 0; o.pad = 0;
            nd.push_back(o);
        }
        // list order of the roots is front..back = root 0..nIni-1, i.e. array must hold them reversed
        std::reverse(nd.begin(), nd.end());
    }
    int live = (int)nd.size();
    auto expand = [&](int i, int* nToExpand) {
        OctNode p = nd[i];
        const int mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1), my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
        const int sb = p.flags & OCT_BUF, db = sb ^ 1;
        int t[4] = {0, 0, 0, 0};
        for (int k = 0; k < p.count; k++) {
            uint32_t key = keys[perm[sb][p.begin + k]];
            t[(oct_x(key) < mx ? 0 : 1) | (oct_y(key) < my ? 0 : 2)]++;
        }
        int s[4] = {p.begin, p.begin + t[0], p.begin + t[0] + t[1], p.begin + t[0] + t[1] + t[2]};
        int r[4] = {0, 0, 0, 0};
        for (int k = 0; k < p.count; k++) {
            uint16_t id = perm[sb][p.begin + k];
            uint32_t key = keys[id];
            int c = (oct_x(key) < mx ? 0 : 1) | (oct_y(key) < my ? 0 : 2);
            perm[db][s[c] + r[c]++] = id;
        }
        const int16_t bx0[4] = {p.x0, (int16_t)mx, p.x0, (int16_t)mx}, bx1[4] = {(int16_t)mx, p.x1, (int16_t)mx, p.x1};
        const int16_t by0[4] = {p.y0, p.y0, (int16_t)my, (int16_t)my}, by1[4] = {(int16_t)my, (int16_t)my, p.y1, p.y1};
        for (int c = 0; c < 4; c++) {
            if (!t[c]) continue;
            OctNode o;
            o.x0 = bx0[c]; o.x1 = bx1[c]; o.y0 = by0[c]; o.y1 = by1[c];
            o.begin = (uint16_t)s[c]; o.count = (uint16_t)t[c];
            o.flags = (uint16_t)(db | (t[c] == 1 ? OCT_NOMORE : 0)); o.pad = 0;
            nd.push_back(o);
            live++;
            if (t[c] > 1 && nToExpand) (*nToExpand)++;
        }
        nd[i].flags |= OCT_DEAD;
        live--;
    };
    auto compact = [&](int upto) {      // stable removal of tombstones; returns new index of `upto`
        int w = 0, first_new = 0;
        for (int i = 0; i < (int)nd.size(); i++) {
            if (i == upto) first_new = w;
            if (!(nd[i].flags & OCT_DEAD)) nd[w++] = nd[i];
        }
        if (upto >= (int)nd.size()) first_new = w;
        nd.resize(w);
        return first_new;
    };
    bool finish = false;
    while (!finish) {
        const int prevSize = live;
        const int nn0 = (int)nd.size();
        int nToExpand = 0;
        for (int i = nn0 - 1; i >= 0; i--)
            if (!(nd[i].flags & (OCT_NOMORE | OCT_DEAD))) expand(i, &nToExpand);
        int first_new = compact(nn0);
        if (live >= N || live == prevSize) {
            finish = true;
        } else if (live + nToExpand * 3 > N) {
            while (!finish) {
                const int prev2 = live;
                std::vector<uint32_t> order;
                for (int i = first_new; i < (int)nd.size(); i++)
                    if (nd[i].count > 1) order.push_back(((uint32_t)nd[i].count << 16) | (uint32_t)i);
                std::sort(order.begin(), order.end());
                const int nn1 = (int)nd.size();
                for (int j = (int)order.size() - 1; j >= 0; j--) {
                    expand((int)(order[j] & 0xffff), nullptr);
                    if (live >= N) break;
                }
                first_new = compact(nn1);
                if (live >= N || live == prev2) finish = true;
            }
        }
    }
    out.reserve(nd.size());
    for (int i = (int)nd.size() - 1; i >= 0; i--) {
        const OctNode& o = nd[i];
        const std::vector<uint16_t>& pm = perm[o.flags & OCT_BUF];
        uint32_t best = keys[pm[o.begin]];
        for (int k = 1; k < o.count; k++) {
            uint32_t key = keys[pm[o.begin + k]];
            if (oct_s(key) > oct_s(best)) best = key;
        }
        out.push_back(best);
    }
    return out;
}

} // namespace viorb
