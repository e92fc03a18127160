// oracle/orb_stereo.h — TEST INFRASTRUCTURE ONLY. CPU restatement of Frame::ComputeStereoMatches
// (reference src/Frame.cc:646-820): row-bucketed Hamming search, 11x11 SAD refinement over 11 shifts on the
// un-blurred pyramid level of the left keypoint, parabola sub-pixel fit, disparity -> depth, median-based rejection.
#pragma once
#include "orb_extractor.h"
#include <vector>
namespace ora {
struct StereoResult { std::vector<float> uRight, depth; std::vector<int> best_sad; };   // -1 where unmatched
StereoResult compute_stereo_matches(const std::vector<KeyPoint>& keysL, const std::vector<uint8_t>& descL,
                                    const std::vector<KeyPoint>& keysR, const std::vector<uint8_t>& descR,
                                    const std::vector<Image8>& pyrL, const std::vector<Image8>& pyrR,
                                    const std::vector<float>& scale, const std::vector<float>& inv_scale, float bf, float fx);
}
