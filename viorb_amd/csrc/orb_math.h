// viorb_amd/csrc/orb_math.h — scalar math shared by the HIP kernels and the host-side setup code.
// Every function here is written so that host (g++) and device (gfx950) evaluate the SAME sequence of
// individually rounded IEEE-754 operations: no fused contraction (the library is built with
// -ffp-contract=off; fma() appears only where it is explicit), no libm transcendental on the device.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define VIORB_HD __host__ __device__ __forceinline__
#else
#define VIORB_HD inline
#endif

namespace viorb {

// cvRound: round-half-to-even of a float (exact: float -> int through rint).
VIORB_HD int round_half_even(float v) { return (int)rintf(v); }

// cv::fastAtan2 (degrees in [0,360)); what IC_Angle returns, reference src/ORBextractor.cc:103.
// OpenCV 2.4.9+ scalar form: odd 7th-order polynomial on min/max ratio, each op rounded to float.
VIORB_HD float fast_atan2_deg(float y, float x) {
    const float k180pi = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k180pi;
    const float p3 = -0.3258083974640975f * k180pi;
    const float p5 = 0.1555786518463281f * k180pi;
    const float p7 = -0.04432655554792128f * k180pi;
    const float eps = 2.2204460492503131e-16f;          // (float)DBL_EPSILON
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cos/sin of a float angle in radians (0 <= r < ~6.3), each returned as the float nearest to the
// exact value: the reference calls cosf/sinf (src/ORBextractor.cc:112-113). Evaluated in double
// with Cody-Waite reduction by pi/2 and Taylor polynomials with explicit fma (error < 2e-16 before
// the single rounding to float), so host and device agree bit for bit; tests/test_host_math.py
// checks it against glibc cosf/sinf.
VIORB_HD void sincos_f32(float r, float* s_out, float* c_out) {
    const double x = (double)r;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00;   // pi/2, 33 significant bits
    const double pio2_lo = 6.07710050650619224932e-11;   // pi/2 - pio2_hi
    const int k = (int)(x * two_over_pi + 0.5);          // x >= 0 on this path
    const double kd = (double)k;
    double y = fma(-kd, pio2_hi, x);
    y = fma(-kd, pio2_lo, y);
    const double z = y * y;
    // sin(y) = y + y*z*(S1 + z*(S2 + ...)), cos(y) = 1 + z*(C1 + z*(C2 + ...)), |y| <= pi/4 + tiny
    double ps = 2.81145725434552076320e-15;              //  1/17!
    ps = fma(ps, z, -7.64716373181981647590e-13);        // -1/15!
    ps = fma(ps, z, 1.60590438368216145994e-10);         //  1/13!
    ps = fma(ps, z, -2.50521083854417187751e-08);        // -1/11!
    ps = fma(ps, z, 2.75573192239858906526e-06);         //  1/9!
    ps = fma(ps, z, -1.98412698412698412698e-04);        // -1/7!
    ps = fma(ps, z, 8.33333333333333333333e-03);         //  1/5!
    ps = fma(ps, z, -1.66666666666666666667e-01);        // -1/3!
    const double sy = fma(y * z, ps, y);
    double pc = 4.77947733238738529744e-14;              //  1/16!
    pc = fma(pc, z, -1.14707455977297247139e-11);        // -1/14!
    pc = fma(pc, z, 2.08767569878680989792e-09);         //  1/12!
    pc = fma(pc, z, -2.75573192239858906526e-07);        // -1/10!
    pc = fma(pc, z, 2.48015873015873015873e-05);         //  1/8!
    pc = fma(pc, z, -1.38888888888888888889e-03);        // -1/6!
    pc = fma(pc, z, 4.16666666666666666667e-02);         //  1/4!
    pc = fma(pc, z, -0.5);
    const double cy = fma(z, pc, 1.0);
    double s, c;
    switch (k & 3) {
        case 0: s = sy; c = cy; break;
        case 1: s = cy; c = -sy; break;
        case 2: s = -sy; c = -cy; break;
        default: s = -cy; c = sy; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

// logf for MapPoint::PredictScale (reference src/MapPoint.cc:408-424: log(ratio)/mfLogScaleFactor, then ceil).
// Evaluated in double and rounded once so host and device agree; the result only feeds a ceil().
VIORB_HD float viorb_logf(float x) { return (float)log((double)x); }

// 256-bit Hamming distance, reference src/ORBmatcher.cc:1648-1664 (SWAR popcount there).
VIORB_HD int hamming256(const uint32_t* a, const uint32_t* b) {
    int d = 0;
    for (int i = 0; i < 8; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
        d += __popc(a[i] ^ b[i]);
#else
        d += __builtin_popcount(a[i] ^ b[i]);
#endif
    }
    return d;
}

} // namespace viorb
