// fcpp_math.h -- the few transcendental functions of a field's SETUP (rotation angle, its sine and cosine, corner angles, edge
// lengths: MLP:165-192, 244-293), written in plain IEEE-754 double operations so that the host (g++/clang, fcpp_host.cpp) and the
// device (gfx950, fcpp_devplan.hip) compute bit-identical values: the setup of a batch runs on the GPU, the same code runs on the
// host as fcpp_plan_count and as the checker of the device-built tables, and both must make the same integer decisions (swath
// counts, reverse-fill counts) and emit the same descriptors.  A platform libm on one side and ocml on the other differ in the last
// bit now and then; the reference itself (numpy's own sin / cos / arctan2) is no more canonical than either.
//
// Compiled with -ffp-contract=off on both sides: a*b+c is two roundings unless fma() is written out; fma() itself is correctly
// rounded on both (hardware on gfx950, libm / hardware on the host).  Accuracy: the rotation's angle, sine and cosine and the edge
// lengths (fc_atan2_cr, fc_sincos_cr, fc_hypot: double-double, round 5) are correctly rounded -- as the platform libm's are in 99.9 % of
// arguments: library, oracle and reference make the same integer decisions even where they hinge on a last bit; the fast forms kept
// for the corner angles are < 1 ulp for sin / cos on |x| <= 1e5, < 1.5 ulp for atan2 (atan2_fd, fcpp_geom.h), ~2 ulp for acos.
// tests/test_shared_math.py measures all of them (the correctly rounded ones against mpmath, bit for bit) through the host library.
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


// ---- round 5: the rotation of a field CORRECTLY ROUNDED.  A field's swath count is int((max_y - min_y) / W) + 1 in the frame the field is
// rotated into (MLP:739): when the height is an exact multiple of the working width the count hinges on the last bit of the rotation's
// angle, sine and cosine -- and the functions above, good to 1 - 1.5 ulp, differ from the platform libm's (the reference's numpy calls it,
// the oracle calls it; both are correctly rounded in 99.9 % of arguments) in 11 % of arguments: 1.9 % of such fields came out a swath apart
// (profiles/r05_fragile_tally.txt).  The angle and its sine / cosine are therefore computed in double-double (error-free sums and
// fma products, ~100 bits) and rounded once: the same bits on host and device as before, and the platform's in all but ~0.1 % of cases.
// Used for the rotation only (once per field; ~800 operations): corner angles and edge lengths keep the functions above.
struct DD { double hi, lo; };
FCPP_HD DD dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; return { s, (a - (s - bb)) + (b - bb) }; }
FCPP_HD DD dd_quick(double a, double b) { const double s = a + b; return { s, b - (s - a) }; }          // |a| >= |b|
FCPP_HD DD dd_two_prod(double a, double b) { const double p = a * b; return { p, fma(a, b, -p) }; }
FCPP_HD DD dd_add(DD a, DD b)
{
    DD s = dd_two_sum(a.hi, b.hi);
    const DD t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick(s.hi, s.lo);
}
FCPP_HD DD dd_mul(DD a, DD b)
{
    DD p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick(p.hi, p.lo);
}
FCPP_HD DD dd_mul_d(DD a, double b)
{
    DD p = dd_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return dd_quick(p.hi, p.lo);
}
// sin and cos of x, |x| <= ~1e5, as double-doubles: reduction by pi/2 given to 160 bits, Taylor series of the remainder (|r| <= pi/4:
// the terms beyond r^27 / 27! and r^28 / 28! are below 1e-33)
FCPP_HD void sincos_dd(double x, DD &s, DD &c)
{
    const double n = rint(x * 6.36619772367581382433e-01);
    const DD p1 = dd_two_prod(n, 1.5707963267948966), p2 = dd_two_prod(n, 6.123233995736766e-17);
    DD r = dd_two_sum(x, -p1.hi);
    r = dd_add(r, DD{ -p1.lo, 0.0 });
    r = dd_add(r, DD{ -p2.hi, -p2.lo });
    r = dd_add(r, DD{ n * 1.4973849048591698e-33, 0.0 });
    const DD z = dd_mul(r, r);
    DD ps = { -9.183689863795546e-29, -1.4303150396787322e-45 };
    ps = dd_add(dd_mul(ps, z), DD{ 6.446950284384474e-26, -1.9330404233703465e-42 });
    ps = dd_add(dd_mul(ps, z), DD{ -3.868170170630684e-23, 8.843177655482344e-40 });
    ps = dd_add(dd_mul(ps, z), DD{ 1.9572941063391263e-20, -1.3643503830087908e-36 });
    ps = dd_add(dd_mul(ps, z), DD{ -8.22063524662433e-18, -2.2141894119604265e-34 });
    ps = dd_add(dd_mul(ps, z), DD{ 2.8114572543455206e-15, 1.6508842730861433e-31 });
    ps = dd_add(dd_mul(ps, z), DD{ -7.647163731819816e-13, -7.03872877733453e-30 });
    ps = dd_add(dd_mul(ps, z), DD{ 1.6059043836821613e-10, 1.2585294588752098e-26 });
    ps = dd_add(dd_mul(ps, z), DD{ -2.505210838544172e-08, 1.448814070935912e-24 });
    ps = dd_add(dd_mul(ps, z), DD{ 2.7557319223985893e-06, -1.858393274046472e-22 });
    ps = dd_add(dd_mul(ps, z), DD{ -0.0001984126984126984, -1.7209558293420705e-22 });
    ps = dd_add(dd_mul(ps, z), DD{ 0.008333333333333333, 1.1564823173178714e-19 });
    ps = dd_add(dd_mul(ps, z), DD{ -0.16666666666666666, -9.25185853854297e-18 });
    ps = dd_add(dd_mul(ps, z), DD{ 1.0, 0.0 });
    const DD sn = dd_mul(ps, r);
    DD pc = { 3.279889237069838e-30, 1.5117542744029879e-46 };
    pc = dd_add(dd_mul(pc, z), DD{ -2.4795962632247976e-27, 1.2953730964765229e-43 });
    pc = dd_add(dd_mul(pc, z), DD{ 1.6117375710961184e-24, -3.6846573564509766e-41 });
    pc = dd_add(dd_mul(pc, z), DD{ -8.896791392450574e-22, 7.911402614872376e-38 });
    pc = dd_add(dd_mul(pc, z), DD{ 4.110317623312165e-19, 1.4412973378659527e-36 });
    pc = dd_add(dd_mul(pc, z), DD{ -1.5619206968586225e-16, -1.1910679660273754e-32 });
    pc = dd_add(dd_mul(pc, z), DD{ 4.779477332387385e-14, 4.399205485834081e-31 });
    pc = dd_add(dd_mul(pc, z), DD{ -1.1470745597729725e-11, -2.0655512752830745e-28 });
    pc = dd_add(dd_mul(pc, z), DD{ 2.08767569878681e-09, -1.20734505911326e-25 });
    pc = dd_add(dd_mul(pc, z), DD{ -2.755731922398589e-07, -2.3767714622250297e-23 });
    pc = dd_add(dd_mul(pc, z), DD{ 2.48015873015873e-05, 2.1511947866775882e-23 });
    pc = dd_add(dd_mul(pc, z), DD{ -0.001388888888888889, 5.300543954373577e-20 });
    pc = dd_add(dd_mul(pc, z), DD{ 0.041666666666666664, 2.3129646346357427e-18 });
    pc = dd_add(dd_mul(pc, z), DD{ -0.5, 0.0 });
    pc = dd_add(dd_mul(pc, z), DD{ 1.0, 0.0 });
    const DD cs = pc;
    const int q = (int)((long long)n & 3);
    s = (q & 1) ? cs : sn;
    c = (q & 1) ? sn : cs;
    if (q == 2 || q == 3) { s.hi = -s.hi; s.lo = -s.lo; }
    if (q == 1 || q == 2) { c.hi = -c.hi; c.lo = -c.lo; }
}
// sin and cos of x correctly rounded (but for arguments whose value lies within ~1e-30 of a rounding boundary)
FCPP_HD void fc_sincos_cr(double x, double &s, double &c)
{
    if (x == 0.0) { s = x; c = 1.0; return; }              // (an unrotated field stays exactly unrotated)
    DD S, C;
    sincos_dd(x, S, C);
    s = S.hi; c = C.hi;                                    // (normalised: hi = RN(hi + lo))
}
// atan2(y, x) correctly rounded: atan2_fd's value (< 1.5 ulp) and ONE Newton step on its residual, the sine and cosine of the estimate in
// double-double -- a0 + (y cos a0 - x sin a0) / (x cos a0 + y sin a0), the numerator a difference of near-equal double-double products
FCPP_HD double fc_atan2_cr(double y, double x)
{
    const double a0 = atan2_fd(y, x);
    if (y == 0.0 || x == 0.0 || !(fabs(a0) > 0.0)) return a0;            // (the axes: exact values of the estimate)
    DD S, C;
    sincos_dd(a0, S, C);
    const DD num = dd_add(dd_mul_d(C, y), dd_mul_d(S, -x));
    const double den = x * C.hi + y * S.hi;
    return a0 + (num.hi + num.lo) / den;
}

// acos(c) for c in [-1, 1] through the one atan2 of the library: atan2(sqrt((1 - c)(1 + c)), c), < 2 ulp.  The planner compares the corner
// angles of a field with 60 degrees (MLP:1043) and with 90 +- 1 degrees (MLP:224-235); a field drawn with such a corner (c = 0.5 to the
// last bit) must fall on the side the reference's acos puts it, so within 1e-6 degrees of those three values the estimate takes ONE Newton
// step on cos(a) - c with the cosine and sine of the estimate in double-double, which makes it the correctly rounded acos there.
FCPP_HD double fc_acos(double c)
{
    if (c >= 1.0) return 0.0;
    if (c <= -1.0) return 3.14159265358979311600e+00;
    const double a0 = atan2_fd(sqrt((1.0 - c) * (1.0 + c)), c);
    const double deg = a0 * (180.0 / 3.14159265358979311600e+00);
    if (!(fabs(deg - 60.0) < 1e-6 || fabs(deg - 89.0) < 1e-6 || fabs(deg - 91.0) < 1e-6)) return a0;
    DD S, C;
    sincos_dd(a0, S, C);
    const DD num = dd_add(C, DD{ -c, 0.0 });
    return a0 + (num.hi + num.lo) / S.hi;
}

// sqrt(x^2 + y^2) for field-sized operands (no scaling: neither overflow nor underflow can occur for coordinates in metres), correctly
// rounded as the platform's hypot is (round 5: the mitre normals of the inset polygon feed the same fragile swath counts as the rotation):
// the sum of the squares in double-double, its root, one Newton correction from the exact residual
FCPP_HD double fc_hypot(double x, double y)
{
    const DD s = dd_add(dd_two_prod(x, x), dd_two_prod(y, y));
    const double r0 = sqrt(s.hi);
    if (!(r0 > 0.0)) return r0;
    const DD r2 = dd_two_prod(r0, r0);
    const DD e = dd_add(s, DD{ -r2.hi, -r2.lo });
    return r0 + (e.hi + e.lo) / (2.0 * r0);
}

}  // namespace fcpp
