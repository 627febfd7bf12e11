// fcpp_math.h -- the few transcendental functions of a field's SETUP (rotation angle, its sine and cosine, corner angles, edge
// lengths: MLP:165-192, 244-293), written in plain IEEE-754 double operations so that the host (g++/clang, fcpp_host.cpp) and the
// device (gfx950, fcpp_devplan.hip) compute bit-identical values: the setup of a batch runs on the GPU, the same code runs on the
// host as fcpp_plan_count and as the checker of the device-built tables, and both must make the same integer decisions (swath
// counts, reverse-fill counts) and emit the same descriptors.  A platform libm on one side and ocml on the other differ in the last
// bit now and then; the reference itself (numpy's own sin / cos / arctan2) is no more canonical than either.
//
// Compiled with -ffp-contract=off on both sides: a*b+c is two roundings unless fma() is written out; fma() itself is correctly
// rounded on both (hardware on gfx950, libm / hardware on the host).  Accuracy: < 1 ulp for sin / cos on |x| <= 1e5, < 1.5 ulp for
// atan2 (atan2_fd, fcpp_geom.h), ~2 ulp for acos; tests/test_shared_math.py measures them against numpy through the host library.
#pragma once
#include <math.h>

#include "fcpp_geom.h"

namespace fcpp {

// sin and cos of x, |x| <= ~1e5: Cody-Waite reduction by pi/2 in three parts (exact products through fma), then the minimax kernels
// of fdlibm (Sun Microsystems' freely distributable libm: published coefficients) on [-pi/4, pi/4]
FCPP_HD void fc_sincos(double x, double &s, double &c)
{
    const double n = rint(x * 6.36619772367581382433e-01);            // x * 2/pi, to the nearest integer (ties to even on both sides)
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    r = fma(-n, -1.49738490485916983e-33, r);
    const double z = r * r;
    // sin r
    const double v = z * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sn = r + v * (-1.66666666666666324348e-01 + z * ps);
    // cos r
    const double w = z * z;
    const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) +
                      w * w * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11));
    const double hz = 0.5 * z, w1 = 1.0 - hz;
    const double cs = w1 + (((1.0 - w1) - hz) + z * pc);
    const int q = (int)((long long)n & 3);
    s = (q & 1) ? cs : sn;
    c = (q & 1) ? sn : cs;
    if (q == 2 || q == 3) s = -s;
    if (q == 1 || q == 2) c = -c;
}

// acos(c) for c in [-1, 1] through the one atan2 of the library: atan2(sqrt((1 - c)(1 + c)), c)
FCPP_HD double fc_acos(double c)
{
    if (c >= 1.0) return 0.0;
    if (c <= -1.0) return 3.14159265358979311600e+00;
    return atan2_fd(sqrt((1.0 - c) * (1.0 + c)), c);
}

// sqrt(x^2 + y^2) for field-sized operands (no scaling: neither overflow nor underflow can occur for coordinates in metres)
FCPP_HD double fc_hypot(double x, double y) { return sqrt(x * x + y * y); }

}  // namespace fcpp
