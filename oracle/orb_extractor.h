// oracle/orb_extractor.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc, include/ORBextractor.h).
#pragma once
#include "cvprim.h"
#include <vector>

namespace ora {

// Layout-compatible with cv::KeyPoint (28 bytes).
struct KeyPoint {
    float x, y, size, angle, response;
    int octave, class_id;
};

class OrbExtractor {
public:
    // reference src/ORBextractor.cc:410-470
    OrbExtractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    // reference src/ORBextractor.cc:1043-1105. `stride` in bytes. Returns number of keypoints.
    int extract(const uint8_t* img, int w, int h, int stride,
                std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc);

    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;                      // reference stores the float ctor arg in a double
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    int8_t pattern[1024];

    // Stage outputs of the last extract() call, kept for stage-by-stage parity tests.
    std::vector<Image8> pyramid;                         // un-padded levels
    std::vector<Image8> blurred;                         // 7x7 sigma-2 blurred levels
    std::vector<std::vector<KeyPoint>> candidates;       // FAST candidates per level (pre-octree),
                                                         // coordinates relative to (minBorderX, minBorderY)
    std::vector<std::vector<KeyPoint>> level_kps;        // after octree + orientation, level coords

    // reference src/ORBextractor.cc:539-763 (+ DivideNode :481-537)
    std::vector<KeyPoint> distribute_octree(const std::vector<KeyPoint>& keys, int minX, int maxX,
                                            int minY, int maxY, int N) const;
private:
    void compute_pyramid(const uint8_t* img, int w, int h, int stride);
    void compute_keypoints();
};

// reference src/ORBextractor.cc:77-104
float ic_angle(const Image8& img, float ptx, float pty, const std::vector<int>& umax);
// reference src/ORBextractor.cc:108-147
void orb_descriptor(const KeyPoint& kp, const Image8& blurred, const int8_t* pattern, uint8_t* desc);

} // namespace ora
