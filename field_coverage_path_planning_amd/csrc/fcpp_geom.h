// fcpp_geom.h -- point-level geometry shared by host setup and HIP kernels (compiled with
// -ffp-contract=off: the reference's numpy arithmetic rounds every product and sum separately,
// so a*b+c must NOT be fused; fma() is written out where fusing is wanted).
#pragma once
#include <math.h>
#include <stdint.h>

#include "fcpp_fresnel_coeffs.h"

#if defined(__HIPCC__)
#define FCPP_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define FCPP_HD inline
#endif

namespace fcpp {

constexpr double kPi = 3.14159265358979323846;
constexpr double kHalfPi = 1.57079632679489661923;

// atan2(y, x) for finite arguments that are not both zero, without the special-case ladder of a library routine: ONE division and an
// odd polynomial.  Method of fdlibm's atan (Sun Microsystems' freely distributable math library; published algorithm and
// coefficients): t = |y| / |x| is reduced at the break points 7/16, 11/16, 19/16, 39/16 to
//     t' = (2t - 1)/(2 + t), (t - 1)/(t + 1), (t - 1.5)/(1 + 1.5 t), -1/t      with atan t = atan(c) + atan t', c = 1/2, 1, 3/2, inf,
// here formed directly from |y| and |x| -- (2|y| - |x|) / (2|x| + |y|) and so on -- so that the quotient y/x itself is never taken;
// |t'| <= 7/16 then goes through the degree-11 polynomial in t'^2.  Error < 1.5 ulp.  Every kernel takes curvature angles from this
// one function (the turning angles of MLP:513-536 that are too large for the short series of curv_chords_fast), so results do not
// depend on which kernel plans a point.
FCPP_HD double atan2_fd(double y, double x)
{
    const double a = fabs(y), b = fabs(x);
    // class of t = a / b (products instead of the quotient: a point within rounding of a break point may fall on either side, both
    // sides being valid reductions)
    const bool c0 = a < 0.4375 * b, c1 = a < 0.6875 * b, c2 = a < 1.1875 * b, c3 = a < 2.4375 * b;
    double num, den, hi, lo;
    if (c0)      { num = a;            den = b;            hi = 0.0;                          lo = 0.0; }
    else if (c1) { num = 2.0 * a - b;  den = 2.0 * b + a;  hi = 4.63647609000806093515e-01;   lo = 2.26987774529616870924e-17; }
    else if (c2) { num = a - b;        den = a + b;        hi = 7.85398163397448278999e-01;   lo = 3.06161699786838301793e-17; }
    else if (c3) { num = a - 1.5 * b;  den = b + 1.5 * a;  hi = 9.82793723247329054082e-01;   lo = 1.39033110312309984516e-17; }
    else         { num = -b;           den = a;            hi = 1.57079632679489655800e+00;   lo = 6.12323399573676603587e-17; }
    const double t = num / den, z = t * t, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02), 6.66107313738753120669e-02),
                                                  9.09088713343650656196e-02), 1.42857142725034663711e-01), 3.33333333333329318027e-01);
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02), -7.69187620504482999495e-02),
                                        -1.11111104054623557880e-01), -1.99999999998764832476e-01);
    double r = hi - ((t * (s1 + s2) - lo) - t);                 // atan(|y| / |x|) in [0, pi/2]
    if (x < 0.0) r = 3.14159265358979311600e+00 - (r - 1.22464679914735317720e-16);
    return y < 0.0 ? -r : r;
}

// unit-curvature clothoid-arc-clothoid shape for one heading change D (pi or pi/2) and clothoid share f
struct CacShape {
    double D, Lc, La, T;  // heading change, clothoid length, arc length, total (all at kappa_max = 1)
    double a;             // sqrt(pi * Lc): Fresnel scaling
    double x1, y1, th1;   // end of the entry clothoid
    double cx, cy;        // arc centre
    double ex, ey;        // end point of the whole turn
    double cosD, sinD;
};

// numpy.linspace(a, b, n)[k] with step = (b - a)/(n - 1) precomputed (IEEE division, same on host)
FCPP_HD double linspace_at(double a, double b, double step, int64_t n, int64_t k)
{
    if (n > 1 && k == n - 1) return b;
    if (step == 0.0) {
        if (b == a) return a;       // (k / (n-1)) * 0 + a: every axis-aligned straight has one coordinate like this
        if (n <= 1) return 0.0 * (b - a) + a;
        return ((double)k / (double)(n - 1)) * (b - a) + a;
    }
    return (double)k * step + a;
}
// the same for 32-bit counts (device code: int -> double is one instruction, int64 -> double is a sequence)
FCPP_HD double linspace_at32(double a, double b, double step, int n, int k)
{
    if (n > 1 && k == n - 1) return b;
    if (step == 0.0) {
        if (b == a) return a;
        if (n <= 1) return 0.0 * (b - a) + a;
        return ((double)k / (double)(n - 1)) * (b - a) + a;
    }
    return (double)k * step + a;
}

namespace detail {
constexpr double kCser[FCPP_FRESNEL_NSER] = FCPP_FRESNEL_CSER_INIT;
constexpr double kSser[FCPP_FRESNEL_NSER] = FCPP_FRESNEL_SSER_INIT;
constexpr double kFcheb[FCPP_FRESNEL_NCHEB] = FCPP_FRESNEL_FCHEB_INIT;
constexpr double kGcheb[FCPP_FRESNEL_NCHEB] = FCPP_FRESNEL_GCHEB_INIT;
}  // namespace detail

// Fresnel integrals C(t) = int_0^t cos(pi u^2/2) du, S(t) likewise with sin.
// |t| <= 1.6: Maclaurin series in t^4 (20 terms); beyond: auxiliary functions f, g (Chebyshev in (1.6/t)^4).
FCPP_HD void fresnel_cs(double t, double &C, double &S)
{
    const double at = fabs(t);
    if (at <= FCPP_FRESNEL_T0) {
        const double t2 = t * t, z = t2 * t2;
        double pc = detail::kCser[FCPP_FRESNEL_NSER - 1], ps = detail::kSser[FCPP_FRESNEL_NSER - 1];
#pragma unroll
        for (int i = FCPP_FRESNEL_NSER - 2; i >= 0; --i) {
            pc = fma(pc, z, detail::kCser[i]);
            ps = fma(ps, z, detail::kSser[i]);
        }
        C = t * pc;
        S = t * t2 * ps;
        return;
    }
    const double q = FCPP_FRESNEL_T0 / at, q2 = q * q, y = q2 * q2, x2 = 2.0 * (2.0 * y - 1.0);
    double f1 = 0, f2 = 0, g1 = 0, g2 = 0;  // Clenshaw
#pragma unroll
    for (int i = FCPP_FRESNEL_NCHEB - 1; i >= 1; --i) {
        double f0 = fma(x2, f1, detail::kFcheb[i]) - f2;
        double g0 = fma(x2, g1, detail::kGcheb[i]) - g2;
        f2 = f1; f1 = f0; g2 = g1; g1 = g0;
    }
    const double F = fma(0.5 * x2, f1, detail::kFcheb[0]) - f2;
    const double G = fma(0.5 * x2, g1, detail::kGcheb[0]) - g2;
    const double f = F / (kPi * at), g = G / (kPi * kPi * at * at * at);
    const double tt = at * at, lo = fma(at, at, -tt);
    const double ph = (fmod(tt, 4.0) + lo) * kHalfPi;
    const double sn = sin(ph), cs = cos(ph);
    double c = 0.5 + (f * sn - g * cs), s = 0.5 - (f * cs + g * sn);
    if (t < 0) { c = -c; s = -s; }
    C = c; S = s;
}

// the Maclaurin branch alone: valid for |t| <= 1.6.  Turn shapes only ever need |t| <= sqrt(f*D/pi) <= 1.
FCPP_HD void fresnel_series(double t, double &C, double &S)
{
    const double t2 = t * t, z = t2 * t2;
    double pc = detail::kCser[FCPP_FRESNEL_NSER - 1], ps = detail::kSser[FCPP_FRESNEL_NSER - 1];
#pragma unroll
    for (int i = FCPP_FRESNEL_NSER - 2; i >= 0; --i) {
        pc = fma(pc, z, detail::kCser[i]);
        ps = fma(ps, z, detail::kSser[i]);
    }
    C = t * pc;
    S = t * t2 * ps;
}

FCPP_HD void cac_unit_point(const CacShape &sh, double u, double &X, double &Y)
{
    if (u < 0) u = 0;
    if (u > sh.T) u = sh.T;
    // entry clothoid (u <= Lc) and exit clothoid (mirrored) share ONE Fresnel evaluation
    const bool entry = sh.Lc > 0 && u <= sh.Lc;
    const bool arc = !entry && (u <= sh.Lc + sh.La || sh.Lc == 0);
    if (arc) {
        const double th = sh.th1 + (u - sh.Lc);
        X = sh.cx + sin(th); Y = sh.cy - cos(th);
        return;
    }
    double c, s;
    fresnel_series((entry ? u : (sh.T - u)) / sh.a, c, s);
    const double qx = sh.a * c, qy = sh.a * s;
    if (entry) { X = qx; Y = qy; }
    else {
        X = sh.ex - (sh.cosD * qx + sh.sinD * qy);
        Y = sh.ey - (sh.sinD * qx - sh.cosD * qy);
    }
}

inline CacShape make_cac_shape(double D, double f)
{
    CacShape sh;
    sh.D = D; sh.Lc = f * D; sh.La = (1 - f) * D; sh.T = 2 * sh.Lc + sh.La;
    sh.a = sqrt(kPi * sh.Lc);
    sh.x1 = 0; sh.y1 = 0; sh.th1 = sh.Lc / 2;
    if (sh.Lc > 0) { double c, s; fresnel_cs(sh.Lc / sh.a, c, s); sh.x1 = sh.a * c; sh.y1 = sh.a * s; }
    sh.cx = sh.x1 - sin(sh.th1); sh.cy = sh.y1 + cos(sh.th1);
    const double th2 = sh.th1 + sh.La;
    const double x2 = sh.cx + sin(th2), y2 = sh.cy - cos(th2);
    sh.cosD = cos(D); sh.sinD = sin(D);
    sh.ex = x2 + (sh.cosD * sh.x1 + sh.sinD * sh.y1);
    sh.ey = y2 + (sh.sinD * sh.x1 - sh.cosD * sh.y1);
    return sh;
}

// world point of a CAC turn: start (x0,y0), heading quadrant q (heading = q*pi/2), turn sign sg (+1 CCW)
FCPP_HD void cac_world_point(const CacShape &sh, double x0, double y0, int q, double sg, double Re, double s,
                             double &x, double &y)
{
    double X, Y;
    cac_unit_point(sh, s / Re, X, Y);
    Y *= sg;
    double rx, ry;  // Rot(q*pi/2) * (X, Y), exact
    switch (q & 3) {
        case 0: rx = X; ry = Y; break;
        case 1: rx = -Y; ry = X; break;
        case 2: rx = -X; ry = -Y; break;
        default: rx = Y; ry = -X; break;
    }
    x = x0 + Re * rx;
    y = y0 + Re * ry;
}

// the same from a unit-shape point (X, Y) = cac_unit_point(sh, s / Re) with Y already multiplied by the turn sign: what the setup of a
// field needs of a corner turn's last two samples, whose unit points are the same for every field of a batch
FCPP_HD void cac_world_from_unit(double X, double Y, double x0, double y0, int q, double Re, double &x, double &y)
{
    double rx, ry;
    switch (q & 3) {
        case 0: rx = X; ry = Y; break;
        case 1: rx = -Y; ry = X; break;
        case 2: rx = -X; ry = -Y; break;
        default: rx = Y; ry = -X; break;
    }
    x = x0 + Re * rx;
    y = y0 + Re * ry;
}

// 90-degree corner arc, quadrant formulas MLP:1049-1060 / 1592-1603
FCPP_HD void corner_arc_point(int ci, double cx, double cy, double R, double c, double s, double &x, double &y)
{
    if (ci == 0)      { x = cx + R * (1 - c); y = cy + R * s; }
    else if (ci == 1) { x = cx - R * s;       y = cy + R * (1 - c); }
    else if (ci == 2) { x = cx - R * (1 - c); y = cy - R * s; }
    else              { x = cx + R * s;       y = cy - R * (1 - c); }
}

}  // namespace fcpp
