// oracle/undistort.cpp — TEST INFRASTRUCTURE ONLY. CPU restatement of the undistortion step of the reference's Frame constructor.
//
// OpenCV is not vendored under /root/reference ("parity unpinned" against the binary); this restates OpenCV 2.4.x cvUndistortPoints as
// published: camera matrix and distortion coefficients converted to double, per point
//     x0 = x = (u - cx) * ifx,  y0 = y = (v - cy) * ify          (ifx = 1./fx, ify = 1./fy)
//     5 fixed-point iterations of
//         r2 = x*x + y*y
//         icdist = (1 + ((k[7]*r2 + k[6])*r2 + k[5])*r2) / (1 + ((k[4]*r2 + k[1])*r2 + k[0])*r2)        k = k1 k2 p1 p2 k3 k4 k5 k6
//         deltaX = 2*k[2]*x*y + k[3]*(r2 + 2*x*x),  deltaY = k[2]*(r2 + 2*y*y) + 2*k[3]*x*y
//         x = (x0 - deltaX)*icdist,  y = (y0 - deltaY)*icdist
//     then the 3x3 "RR = P * R" applied in homogeneous form: xx = RR00*x + RR01*y + RR02, yy = RR10*x + RR11*y + RR12,
//     ww = 1./(RR20*x + RR21*y + RR22), out = (float)(xx*ww), (float)(yy*ww).
// The reference passes R = Mat() (identity) and P = mK, so RR = K (a double 3x3 product with the identity: exact) with zeros off the
// pinhole pattern; the zero products are kept because (fx*x + 0*y) + cx is the literal order (0*y is +-0 and cannot change a finite sum).
// Everything is double with no FMA contraction (-ffp-contract=off), float only at the input and the output.
#include "undistort.h"
#include <algorithm>

namespace ora {

void undistort_points(const float* xy, int n, const float* K4, const float* dist5, float* xy_out) {
    const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    double k[8] = {dist5[0], dist5[1], dist5[2], dist5[3], dist5[4], 0, 0, 0};
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
    for (int i = 0; i < n; i++) {
        double x = xy[2 * i], y = xy[2 * i + 1];
        const double x0 = x = (x - cx) * ifx;
        const double y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

// Frame.cc:616-644
void image_bounds(int width, int height, const float* K4, const float* dist5, float* b) {
    if (dist5[0] != 0.0) {
        float mat[8] = {0.0f, 0.0f, (float)width, 0.0f, 0.0f, (float)height, (float)width, (float)height};
        undistort_points(mat, 4, K4, dist5, mat);
        b[0] = std::min(mat[0], mat[4]);      // mnMinX = min(mat(0,0), mat(2,0))
        b[1] = std::max(mat[2], mat[6]);      // mnMaxX = max(mat(1,0), mat(3,0))
        b[2] = std::min(mat[1], mat[3]);      // mnMinY = min(mat(0,1), mat(1,1))
        b[3] = std::max(mat[5], mat[7]);      // mnMaxY = max(mat(2,1), mat(3,1))
    } else {
        b[0] = 0.0f; b[1] = (float)width; b[2] = 0.0f; b[3] = (float)height;
    }
}

} // namespace ora
