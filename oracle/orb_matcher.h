// oracle/orb_matcher.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the frame-side ORBmatcher searches and the Frame grid they run on
// (reference src/ORBmatcher.cc, src/Frame.cc; lines cited per function in orb_matcher.cpp).
#pragma once
#include "orb_extractor.h"
#include <vector>

namespace ora {

enum { FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64, TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30 };

// reference src/ORBmatcher.cc:1648-1664 (SWAR popcount over 8 x 32 bit)
int descriptor_distance(const uint8_t* a, const uint8_t* b);

// The part of ORB_SLAM2::Frame the matchers read: undistorted keypoints, descriptors, image bounds and
// the 64x48 bucket grid (reference src/Frame.cc:180-186, 410-425, 507-572).
struct FrameGrid {
    int N = 0;
    const KeyPoint* kps = nullptr;            // mvKeysUn
    const uint8_t* desc = nullptr;            // mDescriptors, N x 32
    float minX = 0, maxX = 0, minY = 0, maxY = 0, wInv = 0, hInv = 0;
    std::vector<int> grid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    void build(const KeyPoint* k, const uint8_t* d, int n, float min_x, float max_x, float min_y, float max_y);
    bool pos_in_grid(const KeyPoint& kp, int& px, int& py) const;
    std::vector<int> features_in_area(float x, float y, float r, int minLevel = -1, int maxLevel = -1) const;
};

// Pinhole + pose used by the projection searches: Tcw rows as 12 floats (Rcw row-major, then tcw).
struct PoseF { float Rcw[9], tcw[3], fx, fy, cx, cy; };
// x3Dc = Rcw*x3Dw + tcw the way cv::gemm evaluates the reference's cv::Mat expression (float products,
// left-to-right float sums, tcw added last).
void transform_point(const PoseF& T, const float* Pw, float* Pc);

struct LastFramePoint {                    // one entry of LastFrame.mvpMapPoints[i] (+ what is read through it)
    uint8_t has_point, outlier, has_observations;
    float Pw[3];                           // pMP->GetWorldPos()
    const uint8_t* desc;                   // pMP->GetDescriptor()
    int octave; float angle;               // LastFrame.mvKeys[i].octave, mvKeysUn[i].angle
};

// ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono=true), reference
// src/ORBmatcher.cc:1328-1471. cur_match[i2] = index i of the last-frame point assigned to current
// keypoint i2, or -1 (caller passes it filled with -1, as Tracking fills mvpMapPoints with NULL).
// bMono == false (:1346-1349, 1385-1410): the last frame's pose, the baseline mb, mbf and the current frame's mvuRight.
struct StereoSearch { PoseF last; float mb, bf; const float* uright; };
int search_by_projection_frame(const FrameGrid& cur, const PoseF& Tcur, const float* scale_factors,
                               const std::vector<LastFramePoint>& last, float th, bool check_orientation,
                               std::vector<int>& cur_match, const StereoSearch* stereo = nullptr);

// One local map point as Tracking::SearchLocalPoints sees it (reference src/Tracking.cc:1904-1958).
struct LocalPoint {
    uint8_t valid;            // !isBad()
    uint8_t skip;             // mnLastFrameSeen == CurrentFrame.mnId (already matched in this frame)
    uint8_t has_observations; // Observations() > 0 (ownership rule of the search)
    float Pw[3], normal[3];   // GetWorldPos(), GetNormal()
    float min_dist, max_dist; // mfMinDistance, mfMaxDistance (the 0.8 / 1.2 invariance factors are applied inside)
    const uint8_t* desc;      // GetDescriptor()
};
struct FrustumResult { uint8_t in_view; float proj_x, proj_y, view_cos; int level; float proj_xr; };   // proj_xr = mTrackProjXR (Frame.cc:499)
// Frame::isInFrustum(pMP, viewingCosLimit) + MapPoint::PredictScale, reference src/Frame.cc:449-505,
// src/MapPoint.cc:408-424. Ow = -Rcw^T tcw (mOw), log_scale_factor = log(scaleFactor) as float.
FrustumResult is_in_frustum(const PoseF& T, const float* Ow, float min_x, float max_x, float min_y, float max_y,
                            float log_scale_factor, int nlevels, const LocalPoint& p, float viewing_cos_limit, float bf = 0.0f);
// mOw of Frame::UpdatePoseMatrices (src/Frame.cc:441-447) in cv::gemm float order.
void camera_centre(const PoseF& T, float* Ow);

// Tracking::SearchLocalPoints' isInFrustum loop + ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th),
// reference src/ORBmatcher.cc:45-129. cur_owner_obs[i2] = 1 when current keypoint i2 already holds a map
// point with observations (it can then never be taken). match[i2] = index of the local point assigned
// here, or -1. frustum (optional) receives the per-point isInFrustum fields.
int search_local_points(const FrameGrid& cur, const PoseF& T, const float* scale_factors, int nlevels, float log_scale_factor,
                        const std::vector<LocalPoint>& pts, float th, float nnratio, const uint8_t* cur_owner_obs,
                        std::vector<int>& match, std::vector<FrustumResult>* frustum,
                        const float* cur_uright = nullptr, float bf = 0.0f);   // stereo / RGB-D frame: mvuRight + mbf (the gate of ORBmatcher.cc:91-97)

// reference src/ORBmatcher.cc:1602-1643
void compute_three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3);

} // namespace ora
