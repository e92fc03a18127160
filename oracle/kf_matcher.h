// oracle/kf_matcher.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Parity unpinned against the reference's libraries (no
// vectors exist); pinned by literal numpy re-derivations in tests/test_oracle_kf_matcher.py.
// CPU restatement of the key-frame side matchers LocalMapping calls around the local BA (reference src/LocalMapping.cc:1296,
// 1522, 1547): ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-823, CheckDistEpipolarLine :138-156) and
// ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th) (:825-975; KeyFrame::GetFeaturesInArea src/KeyFrame.cc:906-945,
// MapPoint::PredictScale src/MapPoint.cc:392-407).
#pragma once
#include "orb_matcher.h"
#include <map>
#include <vector>

namespace ora {

struct KfView {                              // what the matchers read of a KeyFrame
    int N = 0;
    const KeyPoint* kps = nullptr;           // mvKeysUn
    const uint8_t* desc = nullptr;           // mDescriptors
    const uint8_t* has_point = nullptr;      // GetMapPoint(i) != NULL
    const float* uright = nullptr;           // mvuRight (< 0: monocular)
    const int* node = nullptr;               // FeatureVector node of each feature (-1: absent)
};
// match12[i1] = i2 or -1; returns nmatches. pose2 = Tcw of key frame 2, Cw1 = camera centre of key frame 1, F12 row-major.
int search_for_triangulation(const KfView& k1, const KfView& k2, const float* F12, const float* Cw1, const PoseF& pose2,
                             const float* scale_factors2, const float* level_sigma2_2, bool only_stereo, bool check_orientation,
                             std::vector<int>& match12);

struct FusePoint { uint8_t valid; float Pw[3], normal[3], min_dist, max_dist; const uint8_t* desc; };   // valid = !isBad() && !IsInKeyFrame(pKF)
// best_idx[p] = key-frame feature the point fuses with (bestDist <= TH_LOW) or -1; returns the number of fused points. The
// Replace / AddObservation bookkeeping that follows in the reference is map management done by the caller.
int fuse(const FrameGrid& kf, const float* uright, const PoseF& T, float bf, const float* scale_factors, const float* inv_level_sigma2,
         int nlevels, float log_scale_factor, const std::vector<FusePoint>& pts, float th, std::vector<int>& best_idx);

} // namespace ora
