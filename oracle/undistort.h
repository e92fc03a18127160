// oracle/undistort.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Frame::UndistortKeyPoints / Frame::ComputeImageBounds (reference src/Frame.cc:584-644) and the OpenCV 2.4 primitive they call,
// cv::undistortPoints(src, dst, K, distCoeffs, R = Mat(), P = K) (modules/imgproc/src/undistort.cpp, cvUndistortPoints).
#pragma once
namespace ora {
// xy[n][2] float pixels in, xy_out[n][2] float pixels out (may alias). K4 = fx fy cx cy, dist5 = k1 k2 p1 p2 k3 (Frame::mDistCoef, float).
void undistort_points(const float* xy, int n, const float* K4, const float* dist5, float* xy_out);
// Frame::ComputeImageBounds: bounds4 = mnMinX mnMaxX mnMinY mnMaxY.
void image_bounds(int width, int height, const float* K4, const float* dist5, float* bounds4);
}
