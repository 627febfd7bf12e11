/*
 * fcpp_oracle.c -- TEST INFRASTRUCTURE ONLY (see fcpp_oracle.h).
 *
 * Sequential float64 restatement of the reference's hot path with the
 * reference's operation order.  Citations "MLP:a-b" are line ranges of
 * /root/reference/multi_layer_planner_v3.py, "GA:a-b" of
 * /root/reference/genetic_algorithm_solver.py (reference @ 2025-10-24).
 *
 * Build with -ffp-contract=off: numpy evaluates a*b+c with two roundings.
 */
#include "fcpp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* numpy.linspace(a, b, n) (endpoint=True): i*step + a, last element := b     */
/* ------------------------------------------------------------------------- */
void orc_linspace(double a, double b, int64_t n, double *out)
{
    if (n <= 0) return;
    if (n == 1) { out[0] = 0.0 * (b - a) + a; return; }
    double div = (double)(n - 1);
    double delta = b - a;
    double step = delta / div;
    if (step == 0.0) {
        for (int64_t i = 0; i < n; ++i) out[i] = ((double)i / div) * delta + a;
    } else {
        for (int64_t i = 0; i < n; ++i) out[i] = (double)i * step + a;
    }
    out[n - 1] = b;
}

/* MLP:513-536 */
double orc_curvature(const double *p1, const double *p2, const double *p3)
{
    double dx1 = p2[0] - p1[0], dy1 = p2[1] - p1[1];
    double dx2 = p3[0] - p2[0], dy2 = p3[1] - p2[1];
    double ds1 = sqrt(dx1 * dx1 + dy1 * dy1);
    double ds2 = sqrt(dx2 * dx2 + dy2 * dy2);
    if (ds1 < 1e-6 || ds2 < 1e-6) return 0.0;
    double theta1 = atan2(dy1, dx1);
    double theta2 = atan2(dy2, dx2);
    double dtheta = theta2 - theta1;
    dtheta = atan2(sin(dtheta), cos(dtheta));
    return fabs(2 * dtheta / (ds1 + ds2));
}

/* MLP:538-589: forward (accel) sweep then backward (decel) sweep, km/h in/out */
void orc_smooth_speed_profile(const double *xy, double *v, int64_t n, double a_lon)
{
    if (n < 2) return;
    for (int64_t i = 1; i < n; ++i) {                      /* MLP:558-571 */
        double dx = xy[2 * i] - xy[2 * (i - 1)], dy = xy[2 * i + 1] - xy[2 * (i - 1) + 1];
        double dist = sqrt(dx * dx + dy * dy);
        if (dist < 1e-6) continue;
        double v1 = v[i - 1] / 3.6;
        double vmax_ms = sqrt(v1 * v1 + 2 * a_lon * dist);
        double vmax_kmh = vmax_ms * 3.6;
        if (v[i] > vmax_kmh) v[i] = vmax_kmh;
    }
    for (int64_t i = n - 2; i >= 0; --i) {                 /* MLP:574-587 */
        double dx = xy[2 * (i + 1)] - xy[2 * i], dy = xy[2 * (i + 1) + 1] - xy[2 * i + 1];
        double dist = sqrt(dx * dx + dy * dy);
        if (dist < 1e-6) continue;
        double v2 = v[i + 1] / 3.6;
        double vmax_ms = sqrt(v2 * v2 + 2 * a_lon * dist);
        double vmax_kmh = vmax_ms * 3.6;
        if (v[i] > vmax_kmh) v[i] = vmax_kmh;
    }
}

/* MLP:467-511; returns the number of clamped points ("speed_adjustments") */
int64_t orc_speed_limit(const double *xy, const double *v_in, double *v_out, int64_t n,
                        const orc_vehicle *veh)
{
    memcpy(v_out, v_in, (size_t)n * sizeof(double));
    if (n < 3) return 0;                                   /* MLP:480-481 (no smoothing either) */
    int64_t adj = 0;
    for (int64_t i = 1; i < n - 1; ++i) {
        double kappa = orc_curvature(xy + 2 * (i - 1), xy + 2 * i, xy + 2 * (i + 1));
        if (kappa > 1e-6) {
            double vmax_ms = sqrt(veh->max_lateral_accel / kappa) * veh->safety_factor;
            double vmax_kmh = vmax_ms * 3.6;
            if (v_out[i] > vmax_kmh) { v_out[i] = vmax_kmh; ++adj; }
        }
    }
    orc_smooth_speed_profile(xy, v_out, n, veh->max_longitudinal_accel);
    return adj;
}

/* MLP:1373-1424 */
void orc_verify(const double *xy, const double *v, int64_t n, const orc_vehicle *veh, double *o)
{
    if (n < 3) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 1; return; }
    double maxk = 0, maxa = 0, maxj = 0, prevk = 0;
    int64_t viol = 0;
    int first = 1;
    for (int64_t i = 1; i < n - 1; ++i) {
        double k = orc_curvature(xy + 2 * (i - 1), xy + 2 * i, xy + 2 * (i + 1));
        double vms = v[i] / 3.6;
        double alat = vms * vms * k;
        if (first || k > maxk) maxk = k;
        if (first || alat > maxa) maxa = alat;
        if (alat > veh->max_lateral_accel) ++viol;
        if (!first) { double j = fabs(k - prevk); if (j > maxj) maxj = j; }
        prevk = k;
        first = 0;
    }
    int64_t m = n - 2;
    o[0] = maxk; o[1] = maxa; o[2] = (double)viol;
    o[3] = m > 0 ? ((double)viol / (double)m * 100) : 0;
    o[4] = maxj;
    o[5] = o[3] < 5 ? 1 : 0;
}

/* numpy.sum over a contiguous float64 array = numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum;
 * numpy is a dependency of the reference, not part of /root/reference: restated from its published algorithm, numpy 2.2):
 * fewer than 8 elements are added left to right; up to 128 elements go through eight running sums combined as
 * ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the remainder left to right; longer arrays are split in halves (the first one a multiple
 * of 8) and the halves' sums added.  With it orc_path_length / orc_work_time equal np.sum(...) of MLP:1296 / 1311 bit for bit. */
static double np_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        int64_t i;
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

/* MLP:1290-1296 */
double orc_path_length(const double *xy, int64_t n)
{
    if (n < 2) return 0.0;
    double *d = (double *)malloc((size_t)(n - 1) * sizeof(double));
    for (int64_t i = 1; i < n; ++i) {
        double dx = xy[2 * i] - xy[2 * (i - 1)], dy = xy[2 * i + 1] - xy[2 * (i - 1) + 1];
        d[i - 1] = sqrt(dx * dx + dy * dy);
    }
    double s = np_pairwise_sum(d, n - 1);
    free(d);
    return s;
}

/* MLP:1298-1311 */
double orc_work_time(const double *xy, const double *v, int64_t n)
{
    if (n < 2) return 0.0;
    double *t = (double *)malloc((size_t)(n - 1) * sizeof(double));
    for (int64_t i = 1; i < n; ++i) {
        double dx = xy[2 * i] - xy[2 * (i - 1)], dy = xy[2 * i + 1] - xy[2 * (i - 1) + 1];
        double d = sqrt(dx * dx + dy * dy);
        double avg = (v[i - 1] + v[i]) / 2;
        double ms = avg / 3.6;
        if (!(ms >= 0.1)) ms = 0.1;   /* np.maximum(avg, 0.1) */
        t[i - 1] = d / ms;
    }
    double s = np_pairwise_sum(t, n - 1);
    free(t);
    return s;
}

/* MLP:1013-1022 */
void orc_straight(double x0, double y0, double x1, double y1, int64_t n, double *xy)
{
    double *t = (double *)malloc((size_t)n * sizeof(double));
    orc_linspace(x0, x1, n, t);
    for (int64_t i = 0; i < n; ++i) xy[2 * i] = t[i];
    orc_linspace(y0, y1, n, t);
    for (int64_t i = 0; i < n; ++i) xy[2 * i + 1] = t[i];
    free(t);
}

/* MLP:791-830 generalised to n points (reference: n = 20) */
static void arc_uturn(double y, int turn_right, double min_x, double max_x, double R, int64_t n,
                      double *xy)
{
    double *ang = (double *)malloc((size_t)n * sizeof(double));
    orc_linspace(0.0, M_PI, n, ang);
    for (int64_t i = 0; i < n; ++i) {
        if (turn_right) {
            xy[2 * i] = max_x - R * cos(ang[i]);
            xy[2 * i + 1] = y + R * sin(ang[i]);
        } else {
            xy[2 * i] = min_x + R * cos(ang[i]);
            xy[2 * i + 1] = y + R * sin(ang[i]);
        }
    }
    free(ang);
}

void orc_safe_arc_turn(double y, int turn_right, double min_x, double max_x, double R, double *xy20)
{
    arc_uturn(y, turn_right, min_x, max_x, R, 20, xy20);
}

/* MLP:1580-1608 (and the identical block MLP:1046-1062); reference n = 15 */
void orc_corner_arc(double x, double y, int ci, double R, int n, double *xy)
{
    double *ang = (double *)malloc((size_t)n * sizeof(double));
    orc_linspace(0.0, M_PI / 2, n, ang);
    for (int i = 0; i < n; ++i) {
        double c = cos(ang[i]), s = sin(ang[i]);
        if (ci == 0)      { xy[2 * i] = x + R * (1 - c); xy[2 * i + 1] = y + R * s; }
        else if (ci == 1) { xy[2 * i] = x - R * s;       xy[2 * i + 1] = y + R * (1 - c); }
        else if (ci == 2) { xy[2 * i] = x - R * (1 - c); xy[2 * i + 1] = y - R * s; }
        else              { xy[2 * i] = x + R * s;       xy[2 * i + 1] = y - R * (1 - c); }
    }
    free(ang);
}

/* MLP:265-284 */
void orc_rotate_point(double x, double y, double ang, double cx, double cy, double *out)
{
    x -= cx; y -= cy;
    double ca = cos(ang), sa = sin(ang);
    double xn = x * ca - y * sa;
    double yn = x * sa + y * ca;
    out[0] = xn + cx; out[1] = yn + cy;
}

/* MLP:1220-1288 (field boundary = bbox anchored at the origin, MLP:1241-1242) */
double orc_distance_to_boundary(double x, double y, double dx, double dy, double L, double H, double R)
{
    double best = 0; int have = 0;
    double cand[4]; int nc = 0;
    if (fabs(dx) > 1e-6) { double t = (0 - x) / dx; if (t > 0) cand[nc++] = t; }
    if (fabs(dx) > 1e-6) { double t = (L - x) / dx; if (t > 0) cand[nc++] = t; }
    if (fabs(dy) > 1e-6) { double t = (0 - y) / dy; if (t > 0) cand[nc++] = t; }
    if (fabs(dy) > 1e-6) { double t = (H - y) / dy; if (t > 0) cand[nc++] = t; }
    for (int i = 0; i < nc; ++i) if (!have || cand[i] < best) { best = cand[i]; have = 1; }
    if (!have) return 2.0 * R;
    double maxd = 3.0 * R;
    return best < maxd ? best : maxd;
}

static int64_t n_for_length(double len, double ds)
{
    int64_t n = (int64_t)ceil(len / ds) + 1;
    return n < 2 ? 2 : n;
}

/* MLP:1154-1218.  spacing == 0 -> reference count max(10, int(len/0.5)); >0 -> dense (BUILD-DEFINED) */
int64_t orc_reverse_path(const double *end, const double *second_last, double L, double H, double R,
                         double spacing, double *len_out, double *xy)
{
    double tx = end[0] - second_last[0], ty = end[1] - second_last[1];
    double nrm = sqrt(tx * tx + ty * ty);
    double dx, dy;
    if (nrm > 1e-6) { dx = -tx / nrm; dy = -ty / nrm; }
    else { dx = -1.0; dy = 0.0; }  /* MLP:1195-1206 needs gap.centroid (GEOS); unreachable for arcs */
    double len = orc_distance_to_boundary(end[0], end[1], dx, dy, L, H, R);
    int64_t n;
    if (spacing > 0) n = n_for_length(len, spacing);
    else { n = (int64_t)(len / 0.5); if (n < 10) n = 10; }
    if (len_out) *len_out = len;
    if (xy) {
        double *t = (double *)malloc((size_t)n * sizeof(double));
        orc_linspace(0.0, len, n, t);
        for (int64_t i = 0; i < n; ++i) {
            xy[2 * i] = end[0] + t[i] * dx;
            xy[2 * i + 1] = end[1] + t[i] * dy;
        }
        free(t);
    }
    return n;
}

/* GA:174-181 */
double orc_ga_distance(const int32_t *route, int32_t n, const double *D)
{
    double total = 0;
    for (int32_t i = 0; i < n; ++i) {
        int32_t a = route[i], b = route[(i + 1) % n];
        total += D[(int64_t)a * n + b];
    }
    return total;
}

/* GA:168-172 */
double orc_ga_fitness(const int32_t *route, int32_t n, const double *D)
{
    return 1.0 / (orc_ga_distance(route, n, D) + 1e-6);
}

/* ------------------------------------------------------------------------- */
/* BUILD-DEFINED: Fresnel integrals by composite Gauss-Legendre in long double */
/* ------------------------------------------------------------------------- */
#define GLN 16
static long double gl_x[GLN], gl_w[GLN];
static int gl_ready = 0;

static void gl_init(void)
{
    if (gl_ready) return;
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < GLN; ++i) {
        long double x = cosl(pi * (i + 0.75L) / (GLN + 0.5L));
        long double dp = 1;
        for (int it = 0; it < 100; ++it) {
            long double p0 = 1, p1 = x;
            for (int k = 2; k <= GLN; ++k) {
                long double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
                p0 = p1; p1 = p2;
            }
            dp = GLN * (x * p1 - p0) / (x * x - 1);
            long double dx = p1 / dp;
            x -= dx;
            if (fabsl(dx) < 1e-21L) break;
        }
        gl_x[i] = x;
        gl_w[i] = 2 / ((1 - x * x) * dp * dp);
    }
    gl_ready = 1;
}

void orc_fresnel(double t, double *C, double *S)
{
    gl_init();
    const long double pi = 3.14159265358979323846264338327950288L;
    long double at = fabsl((long double)t);
    int64_t m = (int64_t)ceill(at / 0.125L);
    if (m < 1) m = 1;
    long double h = at / m, c = 0, s = 0;
    for (int64_t p = 0; p < m; ++p) {
        long double a = p * h, mid = a + h / 2;
        for (int i = 0; i < GLN; ++i) {
            long double u = mid + gl_x[i] * h / 2;
            long double ph = pi * u * u / 2;
            c += gl_w[i] * cosl(ph);
            s += gl_w[i] * sinl(ph);
        }
    }
    c *= h / 2; s *= h / 2;
    if (t < 0) { c = -c; s = -s; }
    *C = (double)c; *S = (double)s;
}

/* unit CAC (kappa_max = 1), total heading change D > 0 (CCW), clothoid share f.
 * Entry clothoid length Lc = f*D (turns f*D/2), arc length La = (1-f)*D, exit clothoid Lc. */
static void cac_unit_point(double D, double f, double s, double *X, double *Y)
{
    double Lc = f * D, La = (1 - f) * D, T = 2 * Lc + La;
    double a = sqrt(M_PI * Lc);
    if (s < 0) s = 0;
    if (s > T) s = T;
    if (Lc > 0 && s <= Lc) {
        double c, sn;
        orc_fresnel(s / a, &c, &sn);
        *X = a * c; *Y = a * sn;
        return;
    }
    double x1 = 0, y1 = 0, th1 = Lc / 2;
    if (Lc > 0) { double c, sn; orc_fresnel(Lc / a, &c, &sn); x1 = a * c; y1 = a * sn; }
    double cx = x1 - sin(th1), cy = y1 + cos(th1);
    if (s <= Lc + La || Lc == 0) {
        double th = th1 + (s - Lc);
        *X = cx + sin(th); *Y = cy - cos(th);
        return;
    }
    /* exit clothoid = entry clothoid mirrored about the turn's axis of symmetry */
    double th2 = th1 + La;
    double x2 = cx + sin(th2), y2 = cy - cos(th2);
    double ex = x2 + (cos(D) * x1 + sin(D) * y1);   /* E = P2 + Rot(D) * (x1, -y1) */
    double ey = y2 + (sin(D) * x1 - cos(D) * y1);
    double ss = T - s, c, sn;
    orc_fresnel(ss / a, &c, &sn);
    double qx = a * c, qy = a * sn;
    *X = ex - (cos(D) * qx + sin(D) * qy);          /* P = E - Rot(D) * (qx, -qy) */
    *Y = ey - (sin(D) * qx - cos(D) * qy);
}

double orc_cac_length(double dth, double Re, double f)
{
    return (1 + f) * fabs(dth) * Re;
}

double orc_cac_fit_radius(double dth, double R, double f, int fit)
{
    if (!fit) return R;
    double D = fabs(dth), X, Y;
    cac_unit_point(D, f, (1 + f) * D, &X, &Y);
    double chord_unit = sqrt(X * X + Y * Y);
    return R * (2 * sin(D / 2)) / chord_unit;
}

void orc_cac_point(double x0, double y0, double th0, double dth, double Re, double f, double s,
                   double *out)
{
    double D = fabs(dth), sg = dth < 0 ? -1.0 : 1.0, X, Y;
    cac_unit_point(D, f, s / Re, &X, &Y);
    Y *= sg;
    out[0] = x0 + Re * (cos(th0) * X - sin(th0) * Y);
    out[1] = y0 + Re * (sin(th0) * X + cos(th0) * Y);
}

/* even-odd crossing rule; points exactly on an edge are not specified */
int orc_point_in_polygon(double px, double py, const double *xy, int64_t nv)
{
    int in = 0;
    for (int64_t i = 0, j = nv - 1; i < nv; j = i++) {
        double xi = xy[2 * i], yi = xy[2 * i + 1], xj = xy[2 * j], yj = xy[2 * j + 1];
        if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) in = !in;
    }
    return in;
}

/* Build-defined (no reference code; include/fcpp.h: fcpp_validate): the geofence rule for an arbitrary simple polygon.
 * signed distance s to the boundary: + inside (even-odd), - outside; returns 1 iff s < -tol */
static double orc_poly_dist2(double px, double py, const double *xy, int64_t nv)
{
    double best = HUGE_VAL;
    for (int64_t k = 0, q = nv - 1; k < nv; q = k++) {
        double ax = xy[2 * q], ay = xy[2 * q + 1], bx = xy[2 * k] - ax, by = xy[2 * k + 1] - ay, wx = px - ax, wy = py - ay;
        double len2 = bx * bx + by * by, dot = wx * bx + wy * by;
        double t = len2 > 0.0 ? dot / len2 : 0.0;
        if (t < 0.0) t = 0.0;
        if (t > 1.0) t = 1.0;
        double dx = wx - t * bx, dy = wy - t * by, d2 = dx * dx + dy * dy;
        if (d2 < best) best = d2;
    }
    return best;
}
int orc_outside_polygon(double px, double py, const double *xy, int64_t nv, double tol)
{
    if (nv < 3) return 0;
    int in = orc_point_in_polygon(px, py, xy, nv);
    double d2 = orc_poly_dist2(px, py, xy, nv), t2 = tol * tol;
    if (in) return tol < 0.0 && d2 < t2;
    return tol < 0.0 || d2 > t2;
}

/* MLP:1086-1152 / :1070: is the area of (2R x 2R square at the corner) - (30-point quarter arc buffered by W/2) above 0.1 m^2?
 * 1 yes, 0 no, -1 not decidable without GEOS (its buffer's round parts are polygons inscribed in the exact ones: with >= 8 segments per
 * quarter circle the buffer lies between the exact buffers of radius (W/2) cos(pi/32) and W/2).  Independent of the library's column
 * integration: rigorous bounds from a 2000 x 2000 grid of cells of side h -- a cell whose CENTRE is farther than r + h/sqrt2 from the
 * polyline is uncovered as a whole (lower bound of the gap for radius r), a cell with any uncovered point has its centre farther than
 * r - h/sqrt2 (upper bound). */
static double orc_corner_gap_sampled(double R, double r)
{
    enum { NP = 30, NG = 2000 };
    double px[NP], py[NP];
    for (int k = 0; k < NP; ++k) { double th = (M_PI / 2) * k / (NP - 1); px[k] = R * (1 - cos(th)); py[k] = R * sin(th); }
    const double h = 2 * R / NG, r2 = r * r;
    long uncovered = 0;
    for (int j = 0; j < NG; ++j)
        for (int i = 0; i < NG; ++i) {
            double x = (i + 0.5) * h, y = (j + 0.5) * h;
            int cov = 0;
            for (int k = 0; k + 1 < NP && !cov; ++k) {
                double ex = px[k + 1] - px[k], ey = py[k + 1] - py[k], wx = x - px[k], wy = y - py[k];
                double t = (wx * ex + wy * ey) / (ex * ex + ey * ey);
                if (t < 0) t = 0;
                if (t > 1) t = 1;
                double dx = wx - t * ex, dy = wy - t * ey;
                cov = dx * dx + dy * dy <= r2;
            }
            uncovered += !cov;
        }
    return uncovered * h * h;
}
int orc_corner_gap_decision(double R, double W)
{
    static double cR = -1, cW = -1;
    static int cD = 0;
    double gap_lb = 4 * R * R - (M_PI * R / 2 * W + M_PI * W * W / 4);
    if (gap_lb > 0.1) return 1;
    if (R == cR && W == cW) return cD;
    const double hs = (2 * R / 2000) * 0.70710678118654757;
    double lo = orc_corner_gap_sampled(R, W / 2 + hs), hi = orc_corner_gap_sampled(R, (W / 2) * cos(M_PI / 32) - hs);
    int d = lo > 0.1 ? 1 : (hi < 0.1 ? 0 : -1);
    cR = R; cW = W; cD = d;
    return d;
}

/* convex polygon, either orientation: outside if beyond any edge by more than tol */
int orc_outside_convex(double px, double py, const double *vx, const double *vy, int nv, double tol)
{
    double a2 = 0;
    for (int i = 0; i < nv; ++i) { int j = (i + 1) % nv; a2 += vx[i] * vy[j] - vx[j] * vy[i]; }
    double sg = a2 >= 0 ? 1.0 : -1.0;
    for (int i = 0; i < nv; ++i) {
        int j = (i + 1) % nv;
        double ex = vx[j] - vx[i], ey = vy[j] - vy[i];
        double ln = sqrt(ex * ex + ey * ey);
        double d = sg * (ex * (py - vy[i]) - ey * (px - vx[i])) / ln;  /* >0 inside */
        if (d < -tol) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* geometry the reference delegates to Shapely, for convex quadrilaterals only */
/* (identical arithmetic to tools/_shapely_standin.py; GEOS itself unpinned)   */
/* ------------------------------------------------------------------------- */
static double poly_area_centroid(const double *vx, const double *vy, int n, double *cx, double *cy)
{
    double a = 0, sx = 0, sy = 0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1) % n;
        double cr = vx[i] * vy[j] - vx[j] * vy[i];
        a += cr;
        sx += (vx[i] + vx[j]) * cr;
        sy += (vy[i] + vy[j]) * cr;
    }
    a *= 0.5;
    if (fabs(a) < 1e-300) { *cx = vx[0]; *cy = vy[0]; return 0.0; }
    *cx = sx / (6.0 * a); *cy = sy / (6.0 * a);
    return a;
}

/* sharp inset by d; returns 0 if empty */
static int inset_convex(const double *vx, const double *vy, int n, double d, double *ox, double *oy)
{
    double cx, cy;
    double a = poly_area_centroid(vx, vy, n, &cx, &cy);
    double sgn = a > 0 ? 1.0 : -1.0;
    double nx[8], ny[8];
    for (int i = 0; i < n; ++i) {
        int j = (i + 1) % n;
        double ex = vx[j] - vx[i], ey = vy[j] - vy[i];
        double ln = hypot(ex, ey);
        nx[i] = -ey / ln * sgn; ny[i] = ex / ln * sgn;
    }
    for (int i = 0; i < n; ++i) {
        int p = (i + n - 1) % n;
        double den = 1.0 + (nx[p] * nx[i] + ny[p] * ny[i]);
        ox[i] = vx[i] + d * (nx[p] + nx[i]) / den;
        oy[i] = vy[i] + d * (ny[p] + ny[i]) / den;
    }
    for (int i = 0; i < n; ++i) {
        int j = (i + 1) % n;
        double ex = vx[j] - vx[i], ey = vy[j] - vy[i];
        if ((ox[j] - ox[i]) * ex + (oy[j] - oy[i]) * ey <= 0) return 0;
    }
    return 1;
}

static double poly_abs_area(const double *vx, const double *vy, int n)
{
    double cx, cy;
    return fabs(poly_area_centroid(vx, vy, n, &cx, &cy));
}

/* MLP:165-192 */
static double corner_angle(const double *vx, const double *vy, int n, int i)
{
    int p = (i + n - 1) % n, q = (i + 1) % n;
    double v1x = vx[p] - vx[i], v1y = vy[p] - vy[i];
    double v2x = vx[q] - vx[i], v2y = vy[q] - vy[i];
    double dot = v1x * v2x + v1y * v2y;
    double n1 = sqrt(v1x * v1x + v1y * v1y), n2 = sqrt(v2x * v2x + v2y * v2y);
    double c = dot / (n1 * n2);
    if (c < -1.0) c = -1.0;
    if (c > 1.0) c = 1.0;
    return acos(c) * (180.0 / M_PI);
}

/* MLP:194-222 */
static int is_parallelogram(const double *vx, const double *vy)
{
    double ex[4], ey[4];
    for (int i = 0; i < 4; ++i) { int j = (i + 1) % 4; ex[i] = vx[j] - vx[i]; ey[i] = vy[j] - vy[i]; }
    int ok = 1;
    for (int k = 0; k < 2; ++k) {
        int a = k, b = k + 2;
        double cross = fabs(ex[a] * ey[b] - ey[a] * ex[b]);
        double na = sqrt(ex[a] * ex[a] + ey[a] * ey[a]), nb = sqrt(ex[b] * ex[b] + ey[b] * ey[b]);
        if (!(cross < 0.01 * (na * nb))) ok = 0;
    }
    return ok;
}

/* growable AoS path with speeds and flag words */
typedef struct { double *xy, *v; uint32_t *fs; int64_t n, cap; } pbuf;
static void pb_reserve(pbuf *b, int64_t extra)
{
    if (b->n + extra <= b->cap) return;
    int64_t c = b->cap ? b->cap : 1024;
    while (c < b->n + extra) c *= 2;
    b->xy = (double *)realloc(b->xy, (size_t)c * 2 * sizeof(double));
    b->v = (double *)realloc(b->v, (size_t)c * sizeof(double));
    b->fs = (uint32_t *)realloc(b->fs, (size_t)c * sizeof(uint32_t));
    b->cap = c;
}
static void pb_push(pbuf *b, const double *xy, int64_t n, double v, uint32_t fs)
{
    pb_reserve(b, n);
    memcpy(b->xy + 2 * b->n, xy, (size_t)n * 2 * sizeof(double));
    for (int64_t i = 0; i < n; ++i) { b->v[b->n + i] = v; b->fs[b->n + i] = fs; }
    b->n += n;
}

/* a turn primitive of either model, n points (BUILD-DEFINED for model 1 / n != reference) */
static void cac_sample(double x0, double y0, double th0, double dth, double R, const orc_options *o,
                       int64_t n, double *xy)
{
    double Re = orc_cac_fit_radius(dth, R, o->clothoid_frac, o->clothoid_fit);
    double T = orc_cac_length(dth, Re, o->clothoid_frac);
    double *s = (double *)malloc((size_t)n * sizeof(double));
    orc_linspace(0.0, T, n, s);
    for (int64_t i = 0; i < n; ++i) orc_cac_point(x0, y0, th0, dth, Re, o->clothoid_frac, s[i], xy + 2 * i);
    free(s);
}
static double turn_length(double dth, double R, const orc_options *o)
{
    if (o->turn_model == 0) return fabs(dth) * R;
    double Re = orc_cac_fit_radius(dth, R, o->clothoid_frac, o->clothoid_fit);
    return orc_cac_length(dth, Re, o->clothoid_frac);
}

/* ---- MLP:599-609 + 288, 696, 710: with obstacles the work area is main_boundary.difference(unary_union(obs.buffer(W / 2))) and the
 * rotation centre of a rotated field is THAT polygon's centroid.  GEOS is not available here; restated for the case that can be written
 * down exactly: every obstacle convex, its buffer inside the main boundary, the grown bounding boxes pairwise disjoint -- then
 *     centroid = (A_b c_b - sum A_k c_k) / (A_b - sum A_k)
 * with A_k, c_k of GEOS' POLYGONAL buffer of obstacle k: the polygon, a W/2-wide rectangle on every edge, and at every vertex a fan of
 * nSegs = int(theta / (pi / 32) + 0.5) equal triangles over the exterior angle theta (Shapely's default 16 segments per quadrant;
 * OffsetSegmentGenerator::addDirectedFillet).  Any other case: the main boundary's own centroid (returns 0) -- parity unpinned either way. */
int orc_difference_centroid(const double *mx, const double *my, const orc_field *f, double r, double *cx, double *cy)
{
    double cbx, cby, Ab = fabs(poly_area_centroid(mx, my, 4, &cbx, &cby));
    int nb = f->n_obstacles;
    if (nb <= 0) return 0;
    double area2 = 0.0;
    for (int i = 0; i < 4; ++i) { int j = (i + 1) % 4; area2 += mx[i] * my[j] - mx[j] * my[i]; }
    double sgn = area2 > 0 ? 1.0 : -1.0;
    double *bb = (double *)malloc((size_t)nb * 4 * sizeof(double));
    double sumA = 0.0, sumAx = 0.0, sumAy = 0.0;
    int ok = 1, nk = 0;
    const double quantum = M_PI / 2 / 16;
    for (int k = 0; k < nb && ok; ++k) {
        int64_t a0 = f->obs_offsets[k], a1 = f->obs_offsets[k + 1];
        int n = (int)(a1 - a0);
        if (n <= 0) continue;
        if (n < 3) { ok = 0; break; }
        const double *q = f->obs_xy + 2 * a0;
        double a2 = 0.0;
        for (int i = 0; i < n; ++i) { int j = (i + 1) % n; a2 += q[2 * i] * q[2 * j + 1] - q[2 * j] * q[2 * i + 1]; }
        if (a2 == 0.0) { ok = 0; break; }
        int rev = a2 < 0;                                   /* walk counter-clockwise */
#define OBV(i, c) q[2 * (rev ? (n - 1 - ((i) % n)) : ((i) % n)) + (c)]
        double A = 0.0, Ax = 0.0, Ay = 0.0;               /* accumulated area and first moments of the buffered polygon */
        double x0 = HUGE_VAL, y0 = HUGE_VAL, x1 = -HUGE_VAL, y1 = -HUGE_VAL;
        for (int i = 0; i < n && ok; ++i) {
            double px = OBV(i, 0), py = OBV(i, 1), qx = OBV(i + 1, 0), qy = OBV(i + 1, 1), sx = OBV(i + 2, 0), sy = OBV(i + 2, 1);
            if (px < x0) x0 = px;
            if (px > x1) x1 = px;
            if (py < y0) y0 = py;
            if (py > y1) y1 = py;
            /* the polygon itself: fan from the origin */
            double cr = px * qy - qx * py;
            A += cr / 2; Ax += cr * (px + qx) / 6; Ay += cr * (py + qy) / 6;
            double ex = qx - px, ey = qy - py, L = sqrt(ex * ex + ey * ey);
            if (!(L > 0)) { ok = 0; break; }
            double nx = ey / L, ny = -ex / L;                /* outward normal of a counter-clockwise polygon */
            /* the rectangle on edge p -> q */
            A += L * r; Ax += L * r * ((px + qx) / 2 + nx * r / 2); Ay += L * r * ((py + qy) / 2 + ny * r / 2);
            /* the fan at vertex q between this edge's normal and the next one's */
            double fx = sx - qx, fy = sy - qy, L2 = sqrt(fx * fx + fy * fy);
            if (!(L2 > 0)) { ok = 0; break; }
            double mx2 = fy / L2, my2 = -fx / L2;
            double crs = nx * my2 - ny * mx2, dt = nx * mx2 + ny * my2;
            if (crs < -1e-12) { ok = 0; break; }              /* not convex */
            double th = atan2(crs < 0 ? 0.0 : crs, dt);
            int ns = (int)(th / quantum + 0.5);
            if (ns < 1) ns = 1;
            double a0n = atan2(ny, nx);
            for (int t = 0; t < ns; ++t) {
                double u0 = a0n + th * t / ns, u1 = a0n + th * (t + 1) / ns;
                double ax = qx + r * cos(u0), ay = qy + r * sin(u0), bx = qx + r * cos(u1), by = qy + r * sin(u1);
                double ta = ((ax - qx) * (by - qy) - (bx - qx) * (ay - qy)) / 2;
                A += ta; Ax += ta * (qx + ax + bx) / 3; Ay += ta * (qy + ay + by) / 3;
                /* the buffer must lie inside the main boundary */
                for (int e = 0; e < 4; ++e) {
                    int j = (e + 1) % 4;
                    double gx = mx[j] - mx[e], gy = my[j] - my[e];
                    if (sgn * (gx * (ay - my[e]) - gy * (ax - mx[e])) < 0 || sgn * (gx * (by - my[e]) - gy * (bx - mx[e])) < 0) ok = 0;
                }
            }
        }
#undef OBV
        if (!ok) break;
        bb[4 * nk] = x0 - r; bb[4 * nk + 1] = y0 - r; bb[4 * nk + 2] = x1 + r; bb[4 * nk + 3] = y1 + r;
        for (int o = 0; o < nk; ++o)
            if (bb[4 * o] <= bb[4 * nk + 2] && bb[4 * nk] <= bb[4 * o + 2] && bb[4 * o + 1] <= bb[4 * nk + 3] && bb[4 * nk + 1] <= bb[4 * o + 3]) ok = 0;
        ++nk;
        sumA += A; sumAx += Ax; sumAy += Ay;
    }
    free(bb);
    if (!ok || nk == 0 || !(Ab - sumA > 0.5 * Ab)) return 0;
    *cx = (Ab * cbx - sumAx) / (Ab - sumA);
    *cy = (Ab * cby - sumAy) / (Ab - sumA);
    return 1;
}

/* ---- BUILD-DEFINED (round 4): obstacle-aware headland.  Boxes in the frame of layer 1 (rotated by -rot about (ccx, ccy)). ---- */
static void to_frame2(double *x, double *y, int rotated, double rot, double ccx, double ccy)
{
    if (rotated) { double o[2]; orc_rotate_point(*x, *y, -rot, ccx, ccy, o); *x = o[0]; *y = o[1]; }
}
static void to_world2(double *x, double *y, int rotated, double rot, double ccx, double ccy)
{
    if (rotated) { double o[2]; orc_rotate_point(*x, *y, rot, ccx, ccy, o); *x = o[0]; *y = o[1]; }
}
/* parameter range of a + t d, t in [0, 1], inside the box and the faces it enters / leaves through (0: x0, 1: x1, 2: y0, 3: y1) */
static int seg_box(double x0, double y0, double x1, double y1, double ax, double ay, double dx, double dy, double *t0, double *t1, int *f0, int *f1)
{
    double a[2] = { ax, ay }, d[2] = { dx, dy }, lo[2] = { x0, y0 }, hi[2] = { x1, y1 };
    *t0 = 0.0; *t1 = 1.0; *f0 = *f1 = -1;
    for (int k = 0; k < 2; ++k) {
        if (fabs(d[k]) < 1e-300) { if (!(a[k] > lo[k] && a[k] < hi[k])) return 0; continue; }
        double ta = (lo[k] - a[k]) / d[k], tb = (hi[k] - a[k]) / d[k];
        int fa = 2 * k, fb = 2 * k + 1;
        if (ta > tb) { double t = ta; ta = tb; tb = t; int f = fa; fa = fb; fb = f; }
        if (ta > *t0) { *t0 = ta; *f0 = fa; }
        if (tb < *t1) { *t1 = tb; *f1 = fb; }
    }
    return *t1 - *t0 > 1e-12;
}
static int box_meets_square(double wx, double wy, double half, int nbox, const double *bx0, const double *by0, const double *bx1, const double *by1,
                            int rotated, double rot, double ccx, double ccy)
{
    to_frame2(&wx, &wy, rotated, rot, ccx, ccy);
    for (int k = 0; k < nbox; ++k)
        if (bx0[k] < wx + half && bx1[k] > wx - half && by0[k] < wy + half && by1[k] > wy - half) return 1;
    return 0;
}
static int box_meets_segment(double ax, double ay, double bx, double by, int nbox, const double *bx0, const double *by0, const double *bx1,
                             const double *by1, int rotated, double rot, double ccx, double ccy)
{
    double t0, t1; int f0, f1;
    to_frame2(&ax, &ay, rotated, rot, ccx, ccy); to_frame2(&bx, &by, rotated, rot, ccx, ccy);
    for (int k = 0; k < nbox; ++k)
        if (seg_box(bx0[k], by0[k], bx1[k], by1[k], ax, ay, bx - ax, by - ay, &t0, &t1, &f0, &f1)) return 1;
    return 0;
}
/* a headland straight A -> B (world) cut at every box it crosses and led around it along the box's boundary -- the shorter way whose
 * corners stay at least W/2 inside the field -- as detour legs; the pieces of the straight keep its sample density (len / 19 at the
 * reference's sampling).  0, or -3: a box over an end of the straight, or no way around inside the field. */
static int headland_straight_around(pbuf *pb, double wax, double way, double wbx, double wby, int nbox, const double *bx0, const double *by0,
                                    const double *bx1, const double *by1, int rotated, double rot, double ccx, double ccy, const double *vx,
                                    const double *vy, double W, double ds, const orc_vehicle *veh, uint32_t fs_straight, int64_t ns_plain)
{
    double len_total = sqrt((wbx - wax) * (wbx - wax) + (wby - way) * (wby - way)), step0 = len_total / 19.0;
    double ax = wax, ay = way, bx = wbx, by = wby;
    to_frame2(&ax, &ay, rotated, rot, ccx, ccy); to_frame2(&bx, &by, rotated, rot, ccx, ccy);
    double dx = bx - ax, dy = by - ay;
    double *ht0 = (double *)malloc((size_t)(nbox + 1) * 2 * sizeof(double)), *ht1 = ht0 + nbox + 1;
    int *hf = (int *)malloc((size_t)(nbox + 1) * 3 * sizeof(int)), *hf0 = hf, *hf1 = hf + nbox + 1, *hb = hf + 2 * (nbox + 1);
    int nh = 0;
    for (int k = 0; k < nbox; ++k) {
        double t0, t1; int f0, f1;
        if (seg_box(bx0[k], by0[k], bx1[k], by1[k], ax, ay, dx, dy, &t0, &t1, &f0, &f1)) {
            int j = nh++;
            while (j > 0 && ht0[j - 1] > t0) { ht0[j] = ht0[j - 1]; ht1[j] = ht1[j - 1]; hf0[j] = hf0[j - 1]; hf1[j] = hf1[j - 1]; hb[j] = hb[j - 1]; --j; }
            ht0[j] = t0; ht1[j] = t1; hf0[j] = f0; hf1[j] = f1; hb[j] = k;
        }
    }
    if (nh == 0) {                     /* no box in the way: the straight as the plain mode samples it */
        int64_t ns = ns_plain;
        double *sb = (double *)malloc((size_t)ns * 2 * sizeof(double));
        orc_straight(wax, way, wbx, wby, ns, sb);
        pb_push(pb, sb, ns, veh->max_headland_speed_kmh, fs_straight);
        free(sb); free(ht0); free(hf);
        return 0;
    }
    uint32_t fs_detour = (fs_straight & ~(uint32_t)ORC_KIND_MASK) | ORC_KIND_DETOUR;
    /* orientation of the field polygon, for "at least W/2 inside" */
    double area2 = 0.0;
    for (int i = 0; i < 4; ++i) { int j = (i + 1) % 4; area2 += vx[i] * vy[j] - vx[j] * vy[i]; }
    double sgn = area2 > 0 ? 1.0 : -1.0;
    int rc = 0;
    double px = ax, py = ay;
#define ORC_LEG(X0, Y0, X1, Y1, DET) do { \
        double x0_ = (X0), y0_ = (Y0), x1_ = (X1), y1_ = (Y1); \
        double len_ = sqrt((x1_ - x0_) * (x1_ - x0_) + (y1_ - y0_) * (y1_ - y0_)); \
        int64_t np_; \
        if (ds > 0) np_ = n_for_length(len_, ds); \
        else { np_ = (int64_t)(len_ / ((DET) ? 0.5 : step0)) + 1; if (np_ < 2) np_ = 2; } \
        to_world2(&x0_, &y0_, rotated, rot, ccx, ccy); to_world2(&x1_, &y1_, rotated, rot, ccx, ccy); \
        double *lb_ = (double *)malloc((size_t)np_ * 2 * sizeof(double)); \
        orc_straight(x0_, y0_, x1_, y1_, np_, lb_); \
        pb_push(pb, lb_, np_, (DET) ? veh->headland_turn_speed_kmh : veh->max_headland_speed_kmh, (DET) ? fs_detour : fs_straight); \
        free(lb_); \
    } while (0)
    for (int h = 0; h < nh && rc == 0; ++h) {
        if (!(ht0[h] > 1e-9 && ht1[h] < 1.0 - 1e-9) || hf0[h] < 0 || hf1[h] < 0) { rc = -3; break; }
        int k = hb[h];
        double X0 = bx0[k], Y0 = by0[k], X1 = bx1[k], Y1 = by1[k];
        double e0x = ax + ht0[h] * dx, e0y = ay + ht0[h] * dy, e1x = ax + ht1[h] * dx, e1y = ay + ht1[h] * dy;
        double *ex[2] = { &e0x, &e1x }, *ey[2] = { &e0y, &e1y };
        int ef[2] = { hf0[h], hf1[h] };
        for (int e = 0; e < 2; ++e) {
            if (ef[e] == 0) *ex[e] = X0; else if (ef[e] == 1) *ex[e] = X1; else if (ef[e] == 2) *ey[e] = Y0; else *ey[e] = Y1;
            if (*ex[e] < X0) *ex[e] = X0;
            if (*ex[e] > X1) *ex[e] = X1;
            if (*ey[e] < Y0) *ey[e] = Y0;
            if (*ey[e] > Y1) *ey[e] = Y1;
        }
        double w = X1 - X0, hh = Y1 - Y0, per = 2 * (w + hh), sp[2];
        for (int e = 0; e < 2; ++e)
            sp[e] = ef[e] == 2 ? *ex[e] - X0 : (ef[e] == 1 ? w + (*ey[e] - Y0) : (ef[e] == 3 ? w + hh + (X1 - *ex[e]) : 2 * w + hh + (Y1 - *ey[e])));
        double cs[4] = { 0.0, w, w + hh, 2 * w + hh }, cxs[4] = { X0, X1, X1, X0 }, cys[4] = { Y0, Y0, Y1, Y1 };
        double best_len = HUGE_VAL;
        int best[4], nbest = 0;
        for (int dir = 0; dir < 2; ++dir) {
            double span = dir == 0 ? fmod(sp[1] - sp[0] + per, per) : fmod(sp[0] - sp[1] + per, per);
            double off[4];
            int cc[4], ncc = 0;
            for (int c = 0; c < 4; ++c) {
                double o = dir == 0 ? fmod(cs[c] - sp[0] + per, per) : fmod(sp[0] - cs[c] + per, per);
                if (o > 1e-9 && o < span - 1e-9) {
                    int j = ncc++;
                    while (j > 0 && off[j - 1] > o) { off[j] = off[j - 1]; cc[j] = cc[j - 1]; --j; }
                    off[j] = o; cc[j] = c;
                }
            }
            int ok = 1;
            for (int q = 0; q < ncc; ++q) {
                double wx = cxs[cc[q]], wy = cys[cc[q]];
                to_world2(&wx, &wy, rotated, rot, ccx, ccy);
                for (int i = 0; i < 4; ++i) {
                    int j = (i + 1) % 4;
                    double gx = vx[j] - vx[i], gy = vy[j] - vy[i], ln = sqrt(gx * gx + gy * gy);
                    if (((-gy / ln * sgn) * (wx - vx[i]) + (gx / ln * sgn) * (wy - vy[i])) < W / 2 - 1e-6) ok = 0;
                }
            }
            if (!ok) continue;
            if (span < best_len - 1e-9) { best_len = span; nbest = ncc; for (int q = 0; q < ncc; ++q) best[q] = cc[q]; }
        }
        if (best_len == HUGE_VAL) { rc = -3; break; }
        ORC_LEG(px, py, e0x, e0y, 0);
        double lx = e0x, ly = e0y;
        for (int q = 0; q < nbest; ++q) { ORC_LEG(lx, ly, cxs[best[q]], cys[best[q]], 1); lx = cxs[best[q]]; ly = cys[best[q]]; }
        ORC_LEG(lx, ly, e1x, e1y, 1);
        px = e1x; py = e1y;
    }
    if (rc == 0) ORC_LEG(px, py, bx, by, 0);
#undef ORC_LEG
    free(ht0); free(hf);
    return rc;
}

/* ---- BUILD-DEFINED (round 4): the W/2-grown POLYGON of an obstacle in the frame of layer 1.  Convex hull of the vertices
 * (counter-clockwise, Andrew's monotone chain), every edge moved `half` outwards, neighbours joined at their mitre point, the result
 * clipped to the obstacle's grown bounding box (Sutherland-Hodgman).  -> vertices written to g (capacity 2 n + 8), 0: no polygon. ---- */
static int cmp_xy(const void *a, const void *b)
{
    const double *p = (const double *)a, *q = (const double *)b;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    if (p[1] != q[1]) return p[1] < q[1] ? -1 : 1;
    return 0;
}
static double cross3(const double *o, const double *a, const double *b) { return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]); }
static int grown_polygon(double *pts /* n x 2, sorted in place */, int n, double half, double x0, double y0, double x1, double y1, double *g)
{
    if (n < 3) return 0;
    qsort(pts, (size_t)n, 2 * sizeof(double), cmp_xy);
    double *h = (double *)malloc((size_t)(2 * n + 2) * 2 * sizeof(double));
    int k = 0;
    for (int i = 0; i < n; ++i) {
        while (k >= 2 && cross3(h + 2 * (k - 2), h + 2 * (k - 1), pts + 2 * i) <= 0) --k;
        h[2 * k] = pts[2 * i]; h[2 * k + 1] = pts[2 * i + 1]; ++k;
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
        while (k >= t && cross3(h + 2 * (k - 2), h + 2 * (k - 1), pts + 2 * i) <= 0) --k;
        h[2 * k] = pts[2 * i]; h[2 * k + 1] = pts[2 * i + 1]; ++k;
    }
    int m = k - 1;
    if (m < 3) { free(h); return 0; }
    double *nrm = (double *)malloc((size_t)m * 2 * sizeof(double));
    double *a = (double *)malloc((size_t)(2 * n + 8) * 2 * sizeof(double)), *b = (double *)malloc((size_t)(2 * n + 8) * 2 * sizeof(double));
    int ok = 1, na = 0;
    for (int i = 0; i < m && ok; ++i) {
        int j = (i + 1) % m;
        double dx = h[2 * j] - h[2 * i], dy = h[2 * j + 1] - h[2 * i + 1], ln = sqrt(dx * dx + dy * dy);
        if (!(ln > 0)) ok = 0;
        else { nrm[2 * i] = dy / ln; nrm[2 * i + 1] = -dx / ln; }
    }
    na = 0;
    for (int i = 0; i < m && ok; ++i) {
        int j = (i + m - 1) % m;
        double n0x = nrm[2 * j], n0y = nrm[2 * j + 1], n1x = nrm[2 * i], n1y = nrm[2 * i + 1];
        double den = 1.0 + (n0x * n1x + n0y * n1y);
        if (den >= 0.5) { a[2 * na] = h[2 * i] + half * (n0x + n1x) / den; a[2 * na + 1] = h[2 * i + 1] + half * (n0y + n1y) / den; ++na; }   /* mitre */
        else {
            /* a sharp vertex: a square cap -- the incoming offset line carried `half` beyond the vertex, the outgoing one begun `half`
             * before it (edge direction = outward normal turned by +90 degrees); the chord stays >= half away from the vertex */
            a[2 * na] = h[2 * i] + half * (n0x - n0y); a[2 * na + 1] = h[2 * i + 1] + half * (n0y + n0x); ++na;
            a[2 * na] = h[2 * i] + half * (n1x + n1y); a[2 * na + 1] = h[2 * i + 1] + half * (n1y - n1x); ++na;
        }
    }
    for (int side = 0; side < 4 && ok; ++side) {
        int nb = 0;
        for (int i = 0; i < na; ++i) {
            const double *p = a + 2 * i, *q = a + 2 * ((i + 1) % na);
            /* (a vertex ON the box is inside whichever way its last bit fell) */
            int ip = side == 0 ? p[0] >= x0 - 1e-9 : (side == 1 ? p[0] <= x1 + 1e-9 : (side == 2 ? p[1] >= y0 - 1e-9 : p[1] <= y1 + 1e-9));
            int iq = side == 0 ? q[0] >= x0 - 1e-9 : (side == 1 ? q[0] <= x1 + 1e-9 : (side == 2 ? q[1] >= y0 - 1e-9 : q[1] <= y1 + 1e-9));
            if (ip) { b[2 * nb] = p[0]; b[2 * nb + 1] = p[1]; ++nb; }
            if (ip != iq) {
                if (side < 2) { double xc = side == 0 ? x0 : x1; b[2 * nb] = xc; b[2 * nb + 1] = p[1] + (q[1] - p[1]) * ((xc - p[0]) / (q[0] - p[0])); }
                else { double yc = side == 2 ? y0 : y1; b[2 * nb + 1] = yc; b[2 * nb] = p[0] + (q[0] - p[0]) * ((yc - p[1]) / (q[1] - p[1])); }
                ++nb;
            }
        }
        double *t = a; a = b; b = t;
        na = nb;
        if (na < 3) ok = 0;
    }
    int ng = 0;
    if (ok) {                          /* vertices closer than 1e-7 to their predecessor are one vertex */
        for (int i = 0; i < na; ++i)
            if (ng == 0 || fabs(a[2 * i] - g[2 * (ng - 1)]) + fabs(a[2 * i + 1] - g[2 * (ng - 1) + 1]) > 1e-7) { g[2 * ng] = a[2 * i]; g[2 * ng + 1] = a[2 * i + 1]; ++ng; }
        while (ng > 1 && fabs(g[0] - g[2 * (ng - 1)]) + fabs(g[1] - g[2 * (ng - 1) + 1]) <= 1e-7) --ng;
        if (ng < 3) ng = 0;
    }
    free(h); free(nrm); free(a); free(b);
    return ng;
}
/* where a line at height y meets the polygon g (m vertices, counter-clockwise, convex) and the vertices of the upper / lower chain from
 * the near point to the far point of a pass travelling left or right; 0: the line passes clear of the polygon */
static int polygon_chains(const double *g, int m, double y, int go_left, double *nearx, double *farx, double *upper, int *nu, double *lower, int *nl)
{
    int eu = -1, ed = -1;
    double xu = 0, xd = 0;
    for (int j = 0; j < m; ++j) {
        const double *a = g + 2 * j, *b = g + 2 * ((j + 1) % m);
        if ((a[1] < y) == (b[1] < y)) continue;
        double x = a[0] + (b[0] - a[0]) * ((y - a[1]) / (b[1] - a[1]));
        if (a[1] < y) { eu = j; xu = x; } else { ed = j; xd = x; }
    }
    if (eu < 0 || ed < 0) return 0;
    int cu = 0, cl = 0;
    /* counter-clockwise: after the up-crossing (right side) the upper vertices g[eu + 1 .. ed], after the down-crossing the lower ones */
    for (int j = (eu + 1) % m;; j = (j + 1) % m) { upper[2 * cu] = g[2 * j]; upper[2 * cu + 1] = g[2 * j + 1]; ++cu; if (j == ed) break; }
    for (int j = (ed + 1) % m;; j = (j + 1) % m) { lower[2 * cl] = g[2 * j]; lower[2 * cl + 1] = g[2 * j + 1]; ++cl; if (j == eu) break; }
    if (go_left) {                     /* right to left: upper chain as listed, lower chain reversed */
        *nearx = xu; *farx = xd;
        for (int i = 0; i < cl / 2; ++i) for (int c = 0; c < 2; ++c) { double t = lower[2 * i + c]; lower[2 * i + c] = lower[2 * (cl - 1 - i) + c]; lower[2 * (cl - 1 - i) + c] = t; }
    } else {
        *nearx = xd; *farx = xu;
        for (int i = 0; i < cu / 2; ++i) for (int c = 0; c < 2; ++c) { double t = upper[2 * i + c]; upper[2 * i + c] = upper[2 * (cu - 1 - i) + c]; upper[2 * (cu - 1 - i) + c] = t; }
    }
    *nu = cu; *nl = cl;
    return 1;
}
/* one leg of a pass in obstacle-aware mode: a numpy.linspace run between its end points in FIELD coordinates */
static void avoid_leg(pbuf *pb, double ax, double ay, double bxx, double byy, int is_detour, uint32_t fs, double ds, int rotated, double rot, double ccx,
                      double ccy, const orc_vehicle *veh)
{
    double len = sqrt((bxx - ax) * (bxx - ax) + (byy - ay) * (byy - ay));
    int64_t np;
    if (ds > 0) np = n_for_length(len, ds);
    else if (is_detour) { np = (int64_t)(len / 0.5) + 1; if (np < 2) np = 2; }
    else np = 2;
    if (rotated) {
        double o[2];
        orc_rotate_point(ax, ay, rot, ccx, ccy, o); ax = o[0]; ay = o[1];
        orc_rotate_point(bxx, byy, rot, ccx, ccy, o); bxx = o[0]; byy = o[1];
    }
    double *lb = (double *)malloc((size_t)np * 2 * sizeof(double));
    orc_straight(ax, ay, bxx, byy, np, lb);
    pb_push(pb, lb, np, is_detour ? veh->headland_turn_speed_kmh : veh->max_work_speed_kmh, fs);
    free(lb);
}

/* MLP:720-789 with the reference's 2-point lines / 20-point arcs */
int64_t orc_u_pattern(double min_x, double min_y, double max_x, double max_y, int reverse_order,
                      int start_from_right, const orc_vehicle *veh, double *xy, double *v, int64_t cap)
{
    double R = veh->min_turn_radius, W = veh->working_width;
    double lsx = min_x + R, lex = max_x - R;
    int64_t P = (int64_t)((max_y - min_y) / W) + 1;
    int64_t n = 0;
    for (int64_t idx = 0; idx < P; ++idx) {
        int64_t i = reverse_order ? (P - 1 - idx) : idx;
        double y = min_y + (double)i * W;
        int go_left = start_from_right ? (idx % 2 == 0) : (idx % 2 == 1);
        if (n + 22 > cap) return -1;
        if (go_left) { xy[2 * n] = lex; xy[2 * n + 1] = y; xy[2 * n + 2] = lsx; xy[2 * n + 3] = y; }
        else         { xy[2 * n] = lsx; xy[2 * n + 1] = y; xy[2 * n + 2] = lex; xy[2 * n + 3] = y; }
        v[n] = v[n + 1] = veh->max_work_speed_kmh;
        n += 2;
        if (idx < P - 1) {
            orc_safe_arc_turn(y, !go_left, min_x, max_x, R, xy + 2 * n);
            for (int k = 0; k < 20; ++k) v[n + k] = veh->headland_turn_speed_kmh;
            n += 20;
        }
    }
    return n;
}

static const double CORNER_TH0[4] = { M_PI / 2, M_PI, -M_PI / 2, 0.0 };

int orc_plan_field(const orc_field *f, const orc_vehicle *veh, const orc_options *opt, orc_plan *out)
{
    memset(out, 0, sizeof(*out));
    const double W = veh->working_width, R = veh->min_turn_radius;
    const double ds = opt->sample_spacing;
    const double *vx = f->vx, *vy = f->vy;

    /* ---- __init__, MLP:63-107 ---- */
    double bminx = vx[0], bmaxx = vx[0], bminy = vy[0], bmaxy = vy[0];
    for (int i = 1; i < 4; ++i) {
        if (vx[i] < bminx) bminx = vx[i];
        if (vx[i] > bmaxx) bmaxx = vx[i];
        if (vy[i] < bminy) bminy = vy[i];
        if (vy[i] > bmaxy) bmaxy = vy[i];
    }
    /* MLP:120-126: vertices -> bbox extents; length/width input -> as given (= vx[1], vy[2]) */
    double L = f->from_vertices ? (bmaxx - bminx) : vx[1];
    double H = f->from_vertices ? (bmaxy - bminy) : vy[2];
    out->field_length = L; out->field_width = H;
    int all90 = 1;
    for (int i = 0; i < 4; ++i) {
        out->corner_angles[i] = corner_angle(vx, vy, 4, i);
        if (!(fabs(out->corner_angles[i] - 90) < 1.0)) all90 = 0;
    }
    out->shape = all90 ? 0 : (is_parallelogram(vx, vy) ? 1 : 2);                 /* MLP:137-163 */
    out->headland_width = R;                                                     /* MLP:310 */
    double hw = R;
    int has_start = f->has_start && (0 <= f->start_x && f->start_x <= L && 0 <= f->start_y && f->start_y <= H);
    int has_end = f->has_end && (0 <= f->end_x && f->end_x <= L && 0 <= f->end_y && f->end_y <= H);
    out->start_kept = has_start; out->end_kept = has_end;                        /* MLP:322-343 */

    /* ---- start corner, MLP:345-385, 396-399 ---- */
    int sci = 0;
    if (has_start) {
        double cxs[4] = { hw / 2, L - hw / 2, L - hw / 2, hw / 2 };
        double cys[4] = { hw / 2, hw / 2, H - hw / 2, H - hw / 2 };
        double best = 0;
        for (int i = 0; i < 4; ++i) {
            double dx = cxs[i] - f->start_x, dy = cys[i] - f->start_y;
            double d = sqrt(dx * dx + dy * dy);
            if (i == 0 || d < best) { best = d; sci = i; }
        }
    }
    out->start_corner = sci;

    /* ---- layer 1, MLP:591-718 ---- */
    double mx[4], my[4];
    if (!inset_convex(vx, vy, 4, hw, mx, my) || poly_abs_area(mx, my, 4) < 1.0) return -1; /* MLP:597-598 */
    double rot = atan2(vy[1] - vy[0], vx[1] - vx[0]);                            /* MLP:255-261 */
    out->rotation_angle = rot;
    int rotated = fabs(rot) > 0.01;                                              /* MLP:686 */
    out->rotated = rotated;
    double ccx = 0, ccy = 0, rx[4], ry[4], sx = f->start_x, sy = f->start_y;
    if (rotated) {
        poly_area_centroid(mx, my, 4, &ccx, &ccy);
        orc_difference_centroid(mx, my, f, W / 2, &ccx, &ccy);      /* MLP:599-609: the work area minus the grown obstacles */
        for (int i = 0; i < 4; ++i) {
            double o[2];
            orc_rotate_point(mx[i], my[i], -rot, ccx, ccy, o);
            rx[i] = o[0]; ry[i] = o[1];
        }
        if (has_start) { double o[2]; orc_rotate_point(sx, sy, -rot, ccx, ccy, o); sx = o[0]; sy = o[1]; }
    } else {
        for (int i = 0; i < 4; ++i) { rx[i] = mx[i]; ry[i] = my[i]; }
    }
    double min_x = rx[0], max_x = rx[0], min_y = ry[0], max_y = ry[0];
    for (int i = 1; i < 4; ++i) {
        if (rx[i] < min_x) min_x = rx[i];
        if (rx[i] > max_x) max_x = rx[i];
        if (ry[i] < min_y) min_y = ry[i];
        if (ry[i] > max_y) max_y = ry[i];
    }
    int reverse_order = 0, start_from_right = 0;                                 /* MLP:631-668 */
    if (has_start) {
        if (sy > (min_y + max_y) / 2) reverse_order = 1;
        if (sx > (min_x + max_x) / 2) start_from_right = 1;
    }
    out->reverse_order = reverse_order; out->start_from_right = start_from_right;

    pbuf pb; memset(&pb, 0, sizeof(pb));
    double *bx0 = NULL, *by0 = NULL, *bx1 = NULL, *by1 = NULL;      /* obstacle-aware swaths: the merged grown boxes, frame of layer 1 */
    int nbox = 0;
    if (opt->obstacle_mode == 1) {
        /* ---- BUILD-DEFINED: obstacle-aware swaths (include/fcpp.h, fcpp_options.obstacle_mode).  Every obstacle = the bounding box
         * of its vertices in the frame of layer 1, grown by W/2; a swath line strictly inside a box's y-range is clipped at the box and
         * the box is passed on three straight legs over its nearer side (top or bottom).  Sub-swaths and legs are numpy.linspace runs
         * between their end points in FIELD coordinates (the end points are rotated back, not the samples); U-turns as in the
         * reference (rotated back point by point). ---- */
        double lsx = min_x + R, lex = max_x - R;
        int64_t P = (int64_t)((max_y - min_y) / W) + 1;
        out->n_swaths = (int32_t)P;
        int64_t n_turn = ds > 0 ? n_for_length(turn_length(M_PI, R, opt), ds) : 20;
        int nb = f->n_obstacles;
        bx0 = (double *)malloc((size_t)(nb + 1) * 4 * sizeof(double)); by0 = bx0 + nb + 1; bx1 = by0 + nb + 1; by1 = bx1 + nb + 1;
        int *ord = (int *)malloc((size_t)(nb + 1) * sizeof(int));
        /* per box the W/2-grown polygon of its obstacle (round 4; gn = 0: merged boxes or no polygon -- passed as a box) */
        double **gp = (double **)calloc((size_t)(nb + 1), sizeof(double *));
        int *gn = (int *)calloc((size_t)(nb + 1), sizeof(int)), gmax = 8;
        for (int k = 0; k < nb; ++k) {
            int64_t a0 = f->obs_offsets[k], a1 = f->obs_offsets[k + 1];
            if (a1 <= a0) continue;
            double x0 = HUGE_VAL, y0 = HUGE_VAL, x1 = -HUGE_VAL, y1 = -HUGE_VAL;
            double *pts = (double *)malloc((size_t)(a1 - a0) * 2 * sizeof(double));
            for (int64_t q = a0; q < a1; ++q) {
                double o[2] = { f->obs_xy[2 * q], f->obs_xy[2 * q + 1] };
                if (rotated) orc_rotate_point(o[0], o[1], -rot, ccx, ccy, o);
                if (o[0] < x0) x0 = o[0];
                if (o[0] > x1) x1 = o[0];
                if (o[1] < y0) y0 = o[1];
                if (o[1] > y1) y1 = o[1];
                pts[2 * (q - a0)] = o[0]; pts[2 * (q - a0) + 1] = o[1];
            }
            bx0[nbox] = x0 - W / 2; by0[nbox] = y0 - W / 2; bx1[nbox] = x1 + W / 2; by1[nbox] = y1 + W / 2;
            gp[nbox] = (double *)malloc((size_t)(2 * (a1 - a0) + 8) * 2 * sizeof(double));
            gn[nbox] = grown_polygon(pts, (int)(a1 - a0), W / 2, bx0[nbox], by0[nbox], bx1[nbox], by1[nbox], gp[nbox]);
            if (2 * (int)(a1 - a0) + 8 > gmax) gmax = 2 * (int)(a1 - a0) + 8;
            free(pts);
            ++nbox;
        }
        /* grown boxes that overlap or touch become one box (their bounding box), until no two do (include/fcpp.h) */
        for (int merged = 1; merged;) {
            merged = 0;
            for (int i = 0; i < nbox; ++i)
                for (int j = i + 1; j < nbox;) {
                    if (bx0[i] <= bx1[j] + 1e-9 && bx0[j] <= bx1[i] + 1e-9 && by0[i] <= by1[j] + 1e-9 && by0[j] <= by1[i] + 1e-9) {
                        if (bx0[j] < bx0[i]) bx0[i] = bx0[j];
                        if (by0[j] < by0[i]) by0[i] = by0[j];
                        if (bx1[j] > bx1[i]) bx1[i] = bx1[j];
                        if (by1[j] > by1[i]) by1[i] = by1[j];
                        gn[i] = 0; free(gp[j]);
                        for (int q = j; q + 1 < nbox; ++q) { bx0[q] = bx0[q + 1]; by0[q] = by0[q + 1]; bx1[q] = bx1[q + 1]; by1[q] = by1[q + 1]; gp[q] = gp[q + 1]; gn[q] = gn[q + 1]; }
                        gp[nbox - 1] = NULL;
                        --nbox;
                        merged = 1;
                    } else ++j;
                }
        }
        double lo0 = lsx < lex ? lsx : lex, hi0 = lsx < lex ? lex : lsx;
        int unsupported = 0;
        double *tb = (double *)malloc((size_t)n_turn * 2 * sizeof(double));
        /* End zones (round 4): the turn after a pass starts where its line ends and occupies DX beyond that end and H above the line
         * (the reference's half circle about (max_x, y): 2 R and R; the clothoid turn: the extents of its shape at 257 samples).  A box
         * that meets the turn's zone, or either line within it, moves the turn inwards until the zone is free (again if the moved zone
         * meets another box); both passes end / start there.  The free ends (start of the first pass, end of the last) are not moved. */
        double DX = 2.0 * R, H = R;
        if (opt->turn_model != 0) {
            double *sb = (double *)malloc(257 * 2 * sizeof(double));
            cac_sample(0.0, 0.0, M_PI / 2, -M_PI, R, opt, 257, sb);
            DX = 0.0; H = 0.0;
            for (int q = 0; q < 257; ++q) { if (sb[2 * q] > DX) DX = sb[2 * q]; if (sb[2 * q + 1] > H) H = sb[2 * q + 1]; }
            free(sb);
        }
        double *clip_lo = (double *)malloc((size_t)(P > 0 ? P : 1) * 2 * sizeof(double)), *clip_hi = clip_lo + (P > 0 ? P : 1);
        for (int64_t q = 0; q < P; ++q) { clip_lo[q] = lo0; clip_hi[q] = hi0; }
        if (lsx < lex && nbox > 0)
            for (int64_t idx = 0; idx + 1 < P; ++idx) {
                int64_t ia = reverse_order ? (P - 1 - idx) : idx, ib = reverse_order ? (P - 2 - idx) : idx + 1;
                double ya = min_y + (double)ia * W, yb = min_y + (double)ib * W, ylo = ya < yb ? ya : yb, yhi = ya + H > yb ? ya + H : yb;
                int gl = start_from_right ? (idx % 2 == 0) : (idx % 2 == 1), right = !gl;
                double x = right ? hi0 : lo0;
                for (int moved = 1; moved;) {
                    moved = 0;
                    for (int k = 0; k < nbox; ++k) {
                        if (!(by0[k] < yhi - 1e-9 && by1[k] > ylo + 1e-9)) continue;
                        if (right ? (bx1[k] > x - 1e-9 && bx0[k] < x + DX + 1e-9 && bx0[k] - DX - 1e-6 < x)
                                  : (bx0[k] < x + 1e-9 && bx1[k] > x - DX - 1e-9 && bx1[k] + DX + 1e-6 > x)) {
                            x = right ? bx0[k] - DX - 1e-6 : bx1[k] + DX + 1e-6;
                            moved = 1;
                        }
                    }
                }
                if (right) clip_hi[idx] = clip_hi[idx + 1] = x;
                else clip_lo[idx] = clip_lo[idx + 1] = x;
            }
        for (int64_t idx = 0; idx < P && !unsupported; ++idx) {
            int64_t i = reverse_order ? (P - 1 - idx) : idx;
            double y = min_y + (double)i * W;
            int go_left = start_from_right ? (idx % 2 == 0) : (idx % 2 == 1);
            int ordered = lsx < lex;
            double lo = clip_lo[idx], hi = clip_hi[idx];
            if (ordered && !(hi - lo > 1e-9)) { unsupported = 1; break; }
            double xs = ordered ? (go_left ? hi : lo) : (go_left ? lex : lsx), xe = ordered ? (go_left ? lo : hi) : (go_left ? lsx : lex);
            uint32_t swath = ORC_KIND_SWATH | ((uint32_t)i << ORC_INDEX_SHIFT), detour = ORC_KIND_DETOUR | ((uint32_t)i << ORC_INDEX_SHIFT);
            /* the boxes this line runs into, in travel order */
            int m = 0;
            for (int k = 0; k < nbox; ++k)
                if (by0[k] < y && y < by1[k] && bx1[k] > lo && bx0[k] < hi) ord[m++] = k;
            for (int a = 1; a < m; ++a) {                       /* insertion sort by the near side */
                int k = ord[a], b = a - 1;
                while (b >= 0 && (go_left ? bx1[ord[b]] < bx1[k] : bx0[ord[b]] > bx0[k])) { ord[b + 1] = ord[b]; --b; }
                ord[b + 1] = k;
            }
            /* legs of the pass: (ax, ay) -> (bx, by) in the frame, kind */
            double cur = xs;
            double *upper = (double *)malloc((size_t)gmax * 4 * sizeof(double)), *lower = upper + 2 * gmax;
            for (int a = 0; a < m && !unsupported; ++a) {
                int k = ord[a];
                double nearx = go_left ? bx1[k] : bx0[k], farx = go_left ? bx0[k] : bx1[k];
                if (!(bx0[k] > lo + 1e-9 && bx1[k] < hi - 1e-9) || !(go_left ? nearx < cur - 1e-9 : nearx > cur + 1e-9)) { unsupported = 1; break; }
                if (gn[k] > 0) {
                    /* round 4: along the obstacle's W/2-grown polygon -- the swath is worked up to the polygon, the way around is the shorter of
                     * its upper and lower chain that stays inside the work area's y-range; a line that passes clear of it is not interrupted */
                    double pnear, pfar;
                    int nu, nl;
                    if (!polygon_chains(gp[k], gn[k], y, go_left, &pnear, &pfar, upper, &nu, lower, &nl)) continue;
                    int top_ok = 1, bot_ok = 1;
                    for (int q = 0; q < nu; ++q) if (!(upper[2 * q + 1] <= max_y + 1e-9)) top_ok = 0;
                    for (int q = 0; q < nl; ++q) if (!(lower[2 * q + 1] >= min_y - 1e-9)) bot_ok = 0;
                    if (!top_ok && !bot_ok) { unsupported = 1; break; }
                    double lens[2];
                    for (int w = 0; w < 2; ++w) {
                        const double *c = w == 0 ? upper : lower;
                        int nc = w == 0 ? nu : nl;
                        double l = 0, qx = pnear, qy = y;
                        for (int q = 0; q < nc; ++q) { l += sqrt((c[2 * q] - qx) * (c[2 * q] - qx) + (c[2 * q + 1] - qy) * (c[2 * q + 1] - qy)); qx = c[2 * q]; qy = c[2 * q + 1]; }
                        lens[w] = l + sqrt((pfar - qx) * (pfar - qx) + (y - qy) * (y - qy));
                    }
                    int take_top = top_ok && (!bot_ok || lens[0] <= lens[1]);
                    const double *c = take_top ? upper : lower;
                    int nc = take_top ? nu : nl;
                    avoid_leg(&pb, cur, y, pnear, y, 0, swath, ds, rotated, rot, ccx, ccy, veh);
                    double qx = pnear, qy = y;
                    for (int q = 0; q < nc; ++q) {
                        if (fabs(c[2 * q] - qx) + fabs(c[2 * q + 1] - qy) > 1e-9) avoid_leg(&pb, qx, qy, c[2 * q], c[2 * q + 1], 1, detour, ds, rotated, rot, ccx, ccy, veh);
                        qx = c[2 * q]; qy = c[2 * q + 1];
                    }
                    avoid_leg(&pb, qx, qy, pfar, y, 1, detour, ds, rotated, rot, ccx, ccy, veh);
                    cur = pfar;
                    continue;
                }
                /* the nearer side if it keeps the detour inside the work area's y-range, else the other; neither: refused */
                int top_ok = by1[k] <= max_y + 1e-9, bot_ok = by0[k] >= min_y - 1e-9, want_top = by1[k] - y <= y - by0[k];
                if (!top_ok && !bot_ok) { unsupported = 1; break; }
                double ys = (want_top ? top_ok : !bot_ok) ? by1[k] : by0[k];
                avoid_leg(&pb, cur, y, nearx, y, 0, swath, ds, rotated, rot, ccx, ccy, veh);
                avoid_leg(&pb, nearx, y, nearx, ys, 1, detour, ds, rotated, rot, ccx, ccy, veh);
                avoid_leg(&pb, nearx, ys, farx, ys, 1, detour, ds, rotated, rot, ccx, ccy, veh);
                avoid_leg(&pb, farx, ys, farx, y, 1, detour, ds, rotated, rot, ccx, ccy, veh);
                cur = farx;
            }
            free(upper);
            if (!unsupported) avoid_leg(&pb, cur, y, xe, y, 0, swath, ds, rotated, rot, ccx, ccy, veh);
            if (unsupported) break;
            if (idx < P - 1) {
                int turn_right = !go_left;
                /* the turn starts where the line ends: the field's own end zone, or one moved inwards by a box */
                double xt = ordered ? (turn_right ? hi : lo) : (turn_right ? lex : lsx);
                if (opt->turn_model == 0) arc_uturn(y, turn_right, turn_right ? min_x : xt - R, turn_right ? xt + R : max_x, R, n_turn, tb);
                else cac_sample(xt, y, M_PI / 2, turn_right ? -M_PI : M_PI, R, opt, n_turn, tb);
                if (rotated)
                    for (int64_t q = 0; q < n_turn; ++q) orc_rotate_point(tb[2 * q], tb[2 * q + 1], rot, ccx, ccy, tb + 2 * q);
                pb_push(&pb, tb, n_turn, veh->headland_turn_speed_kmh, ORC_KIND_UTURN | ((uint32_t)i << ORC_INDEX_SHIFT));
            }
        }
        free(tb); free(ord); free(clip_lo);
        for (int k = 0; k <= nb; ++k) free(gp[k]);
        free(gp); free(gn);
        if (unsupported) { free(bx0); free(pb.xy); free(pb.v); free(pb.fs); return -3; }
    } else {
        double lsx = min_x + R, lex = max_x - R;
        int64_t P = (int64_t)((max_y - min_y) / W) + 1;
        out->n_swaths = (int32_t)P;
        int64_t n_line = ds > 0 ? n_for_length(fabs(lex - lsx), ds) : 2;
        int64_t n_turn = ds > 0 ? n_for_length(turn_length(M_PI, R, opt), ds) : 20;
        double *lb = (double *)malloc((size_t)(n_line > n_turn ? n_line : n_turn) * 2 * sizeof(double));
        for (int64_t idx = 0; idx < P; ++idx) {
            int64_t i = reverse_order ? (P - 1 - idx) : idx;
            double y = min_y + (double)i * W;
            int go_left = start_from_right ? (idx % 2 == 0) : (idx % 2 == 1);
            if (go_left) orc_straight(lex, y, lsx, y, n_line, lb);
            else orc_straight(lsx, y, lex, y, n_line, lb);
            if (n_line == 2) { lb[1] = y; lb[3] = y; }
            pb_push(&pb, lb, n_line, veh->max_work_speed_kmh, ORC_KIND_SWATH | ((uint32_t)i << ORC_INDEX_SHIFT));
            if (idx < P - 1) {
                int turn_right = !go_left;
                if (opt->turn_model == 0) arc_uturn(y, turn_right, min_x, max_x, R, n_turn, lb);
                else if (turn_right) cac_sample(max_x - R, y, M_PI / 2, -M_PI, R, opt, n_turn, lb);
                else cac_sample(min_x + R, y, M_PI / 2, M_PI, R, opt, n_turn, lb);
                pb_push(&pb, lb, n_turn, veh->headland_turn_speed_kmh,
                        ORC_KIND_UTURN | ((uint32_t)i << ORC_INDEX_SHIFT));
            }
        }
        free(lb);
        if (rotated)                                                             /* MLP:709-714 */
            for (int64_t i = 0; i < pb.n; ++i)
                orc_rotate_point(pb.xy[2 * i], pb.xy[2 * i + 1], rot, ccx, ccy, pb.xy + 2 * i);
    }
    int64_t n_main = pb.n;
    out->n_main = n_main;

    /* ---- layer 2, MLP:860-1084 ---- */
    int num_loops = (int)ceil(hw / W);                                           /* MLP:916 */
    out->n_loops = num_loops;
    for (int loop = 0; loop < num_loops; ++loop) {
        double offset = W / 2 + loop * W;                                        /* MLP:924 */
        double cx4[4], cy4[4];
        if (!inset_convex(vx, vy, 4, offset, cx4, cy4) || poly_abs_area(cx4, cy4, 4) < 1.0) {
            free(bx0); free(pb.xy); free(pb.v); free(pb.fs);
            return -2;                                                           /* MLP:967-969 then :939 */
        }
        if (opt->ring_order == 1) {       /* the ring lists the corners the other way round from the same first vertex: 0, 3, 2, 1 */
            double t = cx4[1]; cx4[1] = cx4[3]; cx4[3] = t;
            t = cy4[1]; cy4[1] = cy4[3]; cy4[3] = t;
        }
        uint32_t lp = ORC_FLAG_HEADLAND | ((uint32_t)(loop * 8) << ORC_INDEX_SHIFT);
        double p0[2] = { cx4[sci], cy4[sci] };
        pb_push(&pb, p0, 1, veh->max_headland_speed_kmh, ORC_KIND_HEAD_START | lp | ((uint32_t)sci << ORC_INDEX_SHIFT));
        for (int i = 0; i < 4; ++i) {
            int cur = (sci + i) % 4, nxt = (sci + i + 1) % 4;
            double len = hypot(cx4[nxt] - cx4[cur], cy4[nxt] - cy4[cur]);
            int64_t ns = ds > 0 ? n_for_length(len, ds) : 20;
            if (nbox > 0) {
                /* BUILD-DEFINED (round 4): the straight is led around the grown boxes it crosses; the corner turn at its end must be free */
                int rcode = headland_straight_around(&pb, cx4[cur], cy4[cur], cx4[nxt], cy4[nxt], nbox, bx0, by0, bx1, by1, rotated, rot, ccx, ccy,
                                                     vx, vy, W, ds, veh, ORC_KIND_HEAD_STRAIGHT | lp | ((uint32_t)cur << ORC_INDEX_SHIFT), ns);
                if (rcode == 0 && i < 3 && box_meets_square(cx4[nxt], cy4[nxt], 2.0 * R, nbox, bx0, by0, bx1, by1, rotated, rot, ccx, ccy)) rcode = -3;
                if (rcode != 0) { free(bx0); free(pb.xy); free(pb.v); free(pb.fs); return rcode; }
            } else {
            double *sb = (double *)malloc((size_t)ns * 2 * sizeof(double));
            orc_straight(cx4[cur], cy4[cur], cx4[nxt], cy4[nxt], ns, sb);
            pb_push(&pb, sb, ns, veh->max_headland_speed_kmh,
                    ORC_KIND_HEAD_STRAIGHT | lp | ((uint32_t)cur << ORC_INDEX_SHIFT));
            free(sb);
            }
            if (i < 3) {
                int64_t nt = ds > 0 ? n_for_length(turn_length(M_PI / 2, R, opt), ds) : 15;
                double *tb = (double *)malloc((size_t)nt * 2 * sizeof(double));
                if (opt->turn_model == 0) orc_corner_arc(cx4[nxt], cy4[nxt], nxt, R, (int)nt, tb);
                else cac_sample(cx4[nxt], cy4[nxt], CORNER_TH0[nxt], -M_PI / 2, R, opt, nt, tb);
                pb_push(&pb, tb, nt, veh->headland_turn_speed_kmh,
                        ORC_KIND_CORNER | lp | ((uint32_t)nxt << ORC_INDEX_SHIFT));
                /* MLP:1043, 224-242, 1066-1082 */
                int add_rev = (loop == 0) && (out->corner_angles[nxt] >= 60);
                /* gap.area > 0.1 (MLP:1070): certain from the lower bound 4R^2 - (pi R W/2 + pi W^2/4), else from the gap's own area
                 * (orc_corner_gap_decision); a decision GEOS' polygonal buffer leaves open fails the field (unsupported) */
                int gap_yes = add_rev ? orc_corner_gap_decision(R, W) : 0;
                if (add_rev && gap_yes < 0) { free(bx0); free(tb); free(pb.xy); free(pb.v); free(pb.fs); return -3; }
                if (add_rev && gap_yes > 0 && nt >= 2) {
                    double rl;
                    int64_t nr = orc_reverse_path(tb + 2 * (nt - 1), tb + 2 * (nt - 2), L, H, R, ds, &rl, NULL);
                    double *rb = (double *)malloc((size_t)nr * 2 * sizeof(double));
                    orc_reverse_path(tb + 2 * (nt - 1), tb + 2 * (nt - 2), L, H, R, ds, &rl, rb);
                    if (nbox > 0 && box_meets_segment(rb[0], rb[1], rb[2 * (nr - 1)], rb[2 * (nr - 1) + 1], nbox, bx0, by0, bx1, by1, rotated, rot, ccx, ccy)) {
                        free(rb); free(bx0); free(tb); free(pb.xy); free(pb.v); free(pb.fs);
                        return -3;
                    }
                    pb_push(&pb, rb, nr, 2.5, ORC_KIND_REVERSE | lp | ((uint32_t)nxt << ORC_INDEX_SHIFT));
                    out->n_reverse[nxt] = (int32_t)nr;
                    free(rb);
                }
                free(tb);
            }
        }
    }
    free(bx0);
    int64_t N = pb.n, n_head = N - n_main;
    out->n_head = n_head;

    /* ---- stats before clamp, MLP:616-628, 882-895 ---- */
    out->main_len_m = orc_path_length(pb.xy, n_main);
    out->main_time_pre_s = orc_work_time(pb.xy, pb.v, n_main);
    out->head_len_m = orc_path_length(pb.xy + 2 * n_main, n_head);
    out->head_time_pre_s = orc_work_time(pb.xy + 2 * n_main, pb.v + n_main, n_head);

    /* ---- speed plan on the concatenation, MLP:411-431 ---- */
    double *vout = (double *)malloc((size_t)N * sizeof(double));
    out->n_adjusted = orc_speed_limit(pb.xy, pb.v, vout, N, veh);
    out->main_time_s = orc_work_time(pb.xy, vout, n_main);
    out->head_time_s = orc_work_time(pb.xy + 2 * n_main, vout + n_main, n_head);

    /* ---- verifier over the concatenation (as test/test_multi-layer_planner_v3.py:41-43) ---- */
    double ver[6];
    orc_verify(pb.xy, vout, N, veh, ver);
    out->max_kappa = ver[0]; out->max_alat = ver[1]; out->n_viol = (int64_t)ver[2];
    out->viol_rate = ver[3]; out->max_jump = ver[4]; out->pass = (int32_t)ver[5];

    double *kap = (double *)calloc((size_t)N, sizeof(double));
    for (int64_t i = 1; i < N - 1; ++i) {
        kap[i] = orc_curvature(pb.xy + 2 * (i - 1), pb.xy + 2 * i, pb.xy + 2 * (i + 1));
        double vms = vout[i] / 3.6;
        if (vms * vms * kap[i] > veh->max_lateral_accel) pb.fs[i] |= ORC_FLAG_ALAT;
    }
    /* BUILD-DEFINED geofence / obstacle flags */
    for (int64_t i = 0; i < N; ++i) {
        double px = pb.xy[2 * i], py = pb.xy[2 * i + 1];
        if (orc_outside_convex(px, py, vx, vy, 4, opt->geofence_tol)) { pb.fs[i] |= ORC_FLAG_OUTSIDE; out->n_outside++; }
        for (int32_t k = 0; k < f->n_obstacles; ++k) {
            int64_t a = f->obs_offsets[k], b = f->obs_offsets[k + 1];
            if (orc_point_in_polygon(px, py, f->obs_xy + 2 * a, b - a)) {
                pb.fs[i] |= ORC_FLAG_OBSTACLE; out->n_in_obstacle++;
                break;
            }
        }
    }

    /* ---- connectors, MLP:437-447, 1313-1355 ---- */
    if (has_start && n_head > 0) {
        out->has_approach = 1;
        orc_straight(f->start_x, f->start_y, pb.xy[2 * n_main], pb.xy[2 * n_main + 1], 50, out->approach);
    }
    if (has_end && n_head > 0) {
        out->has_departure = 1;
        orc_straight(pb.xy[2 * (N - 1)], pb.xy[2 * (N - 1) + 1], f->end_x, f->end_y, 50, out->departure);
    }
    out->xy = pb.xy; out->v = vout; out->kappa = kap; out->flagseg = pb.fs;
    free(pb.v);
    return 0;
}

void orc_plan_free(orc_plan *p)
{
    free(p->xy); free(p->v); free(p->kappa); free(p->flagseg);
    p->xy = p->v = p->kappa = NULL; p->flagseg = NULL;
}

/* ---- coverage rasterisation (build-defined sampling of MLP:1357-1371, 1426-1509) ------------------------------------ */
static int orc_seg_covers(double ax, double ay, double bx, double by, double X, double Y, double r2, int strict)
{
    const double abx = bx - ax, aby = by - ay, apx = X - ax, apy = Y - ay;
    const double len2 = abx * abx + aby * aby, dot = apx * abx + apy * aby;
    double lhs, rhs = r2;
    if (dot <= 0.0) lhs = apx * apx + apy * apy;
    else if (dot >= len2) { const double bpx = X - bx, bpy = Y - by; lhs = bpx * bpx + bpy * bpy; }
    else { const double cr = abx * apy - aby * apx; lhs = cr * cr; rhs = r2 * len2; }
    return strict ? (lhs < rhs) : (lhs <= rhs);
}

static int orc_line_covers(const double *x, const double *y, int32_t n, double X, double Y, double r2, int strict)
{
    for (int32_t s = 0; s + 1 < n; ++s)
        if (orc_seg_covers(x[s], y[s], x[s + 1], y[s + 1], X, Y, r2, strict)) return 1;
    return 0;
}

void orc_cover_grid(double ox, double oy, double res, double shift, double radius, int32_t nx, int32_t ny, const double *ax,
                    const double *ay, int32_t n_a, const double *bx, const double *by, int32_t n_b, int strict,
                    const double *region, uint8_t *grid, int64_t *counts)
{
    const double r2 = radius * radius;
    counts[0] = counts[1] = counts[2] = 0;
    for (int32_t j = 0; j < ny; ++j) {
        const double Y = oy + ((double)j + shift) * res;
        for (int32_t i = 0; i < nx; ++i) {
            const double X = ox + ((double)i + shift) * res;      /* MLP:1477-1478 with shift = 0 */
            int in = 1;
            if (region) {
                int io = 1, ii = 1;
                for (int e = 0; e < 4; ++e) {
                    io = io && (region[3 * e] * X + region[3 * e + 1] * Y + region[3 * e + 2] >= 0.0);
                    ii = ii && (region[12 + 3 * e] * X + region[12 + 3 * e + 1] * Y + region[12 + 3 * e + 2] >= 0.0);
                }
                in = io && !ii;
            }
            uint8_t g = 0;
            if (in) {
                counts[0]++;
                if (orc_line_covers(ax, ay, n_a, X, Y, r2, strict)) g = 1;                    /* MLP:1481-1483 */
                else if (n_b > 1 && orc_line_covers(bx, by, n_b, X, Y, r2, strict)) g = 2;    /* MLP:1489-1497 */
                counts[1] += (g == 1);
                counts[2] += (g != 0);
            }
            if (grid) grid[(int64_t)j * nx + i] = g;
        }
    }
}

/* ---- GA evolution (GA:64-115, 183-268) ----------------------------------------------------------------------------- */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double orc_unit(uint32_t a, uint32_t b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0; }

void orc_ga_selection(const int32_t *population, const double *fitness, int32_t pop, int32_t n, const int32_t *cand, int32_t k,
                      int32_t *selected)
{
    for (int32_t s = 0; s < pop; ++s) {
        int32_t w = cand[(int64_t)s * k];
        for (int32_t j = 1; j < k; ++j) {                      /* np.argmax: the FIRST maximum wins */
            const int32_t c = cand[(int64_t)s * k + j];
            if (fitness[c] > fitness[w]) w = c;
        }
        memcpy(selected + (int64_t)s * n, population + (int64_t)w * n, (size_t)n * sizeof(int32_t));
    }
}

static void orc_ox_child(const int32_t *keep, const int32_t *donor, int32_t n, int32_t a, int32_t b, int32_t *child)
{
    unsigned char *present = (unsigned char *)calloc((size_t)n, 1);
    for (int32_t i = a; i < b; ++i) { child[i] = keep[i]; present[keep[i]] = 1; }      /* GA:225-227 */
    int32_t pos = b;
    for (int32_t q = 0; q < n; ++q) {                                                  /* parent[cx2:] + parent[:cx2] */
        const int32_t gene = donor[(b + q) % n];
        if (!present[gene]) {
            if (pos >= n) pos = 0;
            child[pos++] = gene;
        }
    }
    free(present);
}

void orc_ga_ox(const int32_t *p1, const int32_t *p2, int32_t n, int32_t a, int32_t b, int32_t *c1, int32_t *c2)
{
    orc_ox_child(p1, p2, n, a, b, c1);
    orc_ox_child(p2, p1, n, a, b, c2);
}

/* the t-th best (t = 0: best) in the order (fitness, index) */
static int32_t orc_next_best(const double *fit, int32_t pop, int have_prev, double pf, int32_t pi)
{
    int32_t best = -1;
    for (int32_t i = 0; i < pop; ++i) {
        if (have_prev && !(fit[i] < pf || (fit[i] == pf && i < pi))) continue;
        if (best < 0 || fit[i] > fit[best] || (fit[i] == fit[best] && i > best)) best = i;
    }
    return best;
}

void orc_ga_elitism(const int32_t *old_population, const double *old_fitness, int32_t pop, int32_t n, int32_t e,
                    int32_t *new_population)
{
    int32_t prev = -1;
    for (int32_t t = 0; t < e; ++t) {
        prev = orc_next_best(old_fitness, pop, t > 0, t > 0 ? old_fitness[prev] : 0.0, prev);
        memcpy(new_population + (int64_t)(pop - 1 - t) * n, old_population + (int64_t)prev * n, (size_t)n * sizeof(int32_t));
    }
}

static void orc_two_positions(uint32_t r0, uint32_t r1, int32_t n, int32_t *i, int32_t *j)
{
    *i = (int32_t)(r0 % (uint32_t)n);
    *j = (int32_t)(r1 % (uint32_t)(n - 1));
    if (*j >= *i) ++*j;
}

/* mean in the summation order of the HIP kernel: 1024 strided partial sums, then a binary tree */
static double orc_tree_mean(const double *v, int32_t m)
{
    double part[1024];
    for (int t = 0; t < 1024; ++t) { double a = 0.0; for (int32_t i = t; i < m; i += 1024) a += v[i]; part[t] = a; }
    for (int o = 512; o > 0; o >>= 1) for (int t = 0; t < o; ++t) part[t] += part[t + o];
    return part[0] / (double)m;
}

void orc_ga_evolve(int32_t n, const orc_ga_config *cfg, const double *D, int32_t *routes, int32_t *best_route, double *hist,
                   orc_ga_result *res)
{
    const int32_t pop = cfg->population_size, k = cfg->tournament_size, e = cfg->elite_size, G = cfg->max_generations;
    const uint32_t key[2] = { (uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32) };
    int32_t *cur = routes, *nxt = (int32_t *)malloc((size_t)pop * n * sizeof(int32_t));
    double *fit = (double *)malloc((size_t)pop * sizeof(double)), *dist = (double *)malloc((size_t)pop * sizeof(double));
    double *nfit = (double *)malloc((size_t)pop * sizeof(double)), *ndist = (double *)malloc((size_t)pop * sizeof(double));
    int32_t *c1 = (int32_t *)malloc((size_t)n * sizeof(int32_t)), *c2 = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    for (int32_t r = 0; r < pop; ++r) { dist[r] = orc_ga_distance(cur + (int64_t)r * n, n, D); fit[r] = 1.0 / (dist[r] + 1e-6); }
    int32_t bi = 0;
    for (int32_t r = 1; r < pop; ++r) if (fit[r] > fit[bi]) bi = r;                 /* GA:66 np.argmax */
    double best_fit = fit[bi], best_dist = dist[bi];
    memcpy(best_route, cur + (int64_t)bi * n, (size_t)n * sizeof(int32_t));
    int32_t gwi = 0, gen = -1;
    for (int32_t g = 0; g < G; ++g) {
        gen = g;
        for (int32_t p = 0; p < pop / 2; ++p) {
            int32_t w[2];
            for (int slot = 0; slot < 2; ++slot) {                                  /* GA:189-194 */
                int32_t cand[64] = { 0 }, nc = 0;
                for (uint32_t j = 0; nc < k; ++j) {
                    const uint32_t ctr[4] = { (uint32_t)g, (uint32_t)p, (uint32_t)(1 + slot), j >> 2 };
                    uint32_t o[4];
                    orc_philox4x32(ctr, key, o);
                    const int32_t c = (int32_t)(o[j & 3] % (uint32_t)pop);
                    int dup = 0;
                    for (int32_t q = 0; q < nc; ++q) dup |= cand[q] == c;
                    if (!dup) cand[nc++] = c;
                }
                w[slot] = cand[0];
                for (int32_t q = 1; q < k; ++q) if (fit[cand[q]] > fit[w[slot]]) w[slot] = cand[q];
            }
            const int32_t *p1 = cur + (int64_t)w[0] * n, *p2 = cur + (int64_t)w[1] * n;
            uint32_t X[4];
            { const uint32_t ctr[4] = { (uint32_t)g, (uint32_t)p, 3u, 0u }; orc_philox4x32(ctr, key, X); }
            if (orc_unit(X[0], X[1]) < cfg->crossover_rate) {                         /* GA:207 */
                int32_t i, j;
                orc_two_positions(X[2], X[3], n, &i, &j);
                orc_ga_ox(p1, p2, n, i < j ? i : j, i < j ? j : i, c1, c2);
            } else { memcpy(c1, p1, (size_t)n * sizeof(int32_t)); memcpy(c2, p2, (size_t)n * sizeof(int32_t)); }
            for (int c = 0; c < 2; ++c) {                                           /* GA:246-250 */
                uint32_t M[4];
                const uint32_t ctr[4] = { (uint32_t)g, (uint32_t)p, (uint32_t)(4 + c), 0u };
                orc_philox4x32(ctr, key, M);
                if (orc_unit(M[0], M[1]) < cfg->mutation_rate) {
                    int32_t i, j, *ch = c ? c2 : c1;
                    orc_two_positions(M[2], M[3], n, &i, &j);
                    const int32_t t = ch[i]; ch[i] = ch[j]; ch[j] = t;
                }
                const int32_t row = 2 * p + c;
                if (row < pop - e) {
                    memcpy(nxt + (int64_t)row * n, c ? c2 : c1, (size_t)n * sizeof(int32_t));
                    ndist[row] = orc_ga_distance(nxt + (int64_t)row * n, n, D);
                    nfit[row] = 1.0 / (ndist[row] + 1e-6);
                }
            }
        }
        {   /* GA:254-268 */
            int32_t prev = -1;
            for (int32_t t = 0; t < e; ++t) {
                prev = orc_next_best(fit, pop, t > 0, t > 0 ? fit[prev] : 0.0, prev);
                memcpy(nxt + (int64_t)(pop - 1 - t) * n, cur + (int64_t)prev * n, (size_t)n * sizeof(int32_t));
                nfit[pop - 1 - t] = fit[prev]; ndist[pop - 1 - t] = dist[prev];
            }
        }
        { int32_t *t = cur; cur = nxt; nxt = t; double *u = fit; fit = nfit; nfit = u; u = dist; dist = ndist; ndist = u; }
        bi = 0;
        for (int32_t r = 1; r < pop; ++r) if (fit[r] > fit[bi]) bi = r;             /* GA:91 */
        if (fit[bi] > best_fit) {                                                   /* GA:94-98 */
            best_fit = fit[bi]; best_dist = dist[bi]; gwi = 0;
            memcpy(best_route, cur + (int64_t)bi * n, (size_t)n * sizeof(int32_t));
        } else ++gwi;
        if (hist) { hist[g] = best_fit; hist[G + g] = orc_tree_mean(fit, pop); }    /* GA:106-107 */
        if (gwi >= cfg->convergence_threshold) break;                               /* GA:110-113 */
    }
    res->generations = gen + 1; res->convergence_gen = gen - gwi;                   /* GA:122-127 */
    res->best_distance = best_dist; res->best_fitness = best_fit;
    if (cur != routes) memcpy(routes, cur, (size_t)pop * n * sizeof(int32_t));
    free(cur == routes ? nxt : cur); free(fit); free(dist); free(nfit); free(ndist); free(c1); free(c2);
}

/* ---- scheduler inputs (MVP:229-259, MFP:263-320) ---------------------------------------------------------------------- */
void orc_distance_matrix(int32_t n, const double *x, const double *y, double *D)
{
    for (int32_t i = 0; i < n; ++i)
        for (int32_t j = 0; j < n; ++j) {
            const double dx = x[i] - x[j], dy = y[i] - y[j];
            D[(int64_t)i * n + j] = i == j ? 0.0 : sqrt(dx * dx + dy * dy);      /* np.linalg.norm of a 2-vector */
        }
}

double orc_best_connection(const double *fx, const double *fy, int64_t nf, const double *tx, const double *ty, int64_t nt,
                           int64_t *bf, int64_t *bt)
{
    double best = INFINITY;
    *bf = *bt = -1;
    for (int64_t a = 0; a < nf; ++a)
        for (int64_t b = 0; b < nt; ++b) {
            const double dx = fx[a] - tx[b], dy = fy[a] - ty[b], d = sqrt(dx * dx + dy * dy);
            if (d < best) { best = d; *bf = a; *bt = b; }                         /* MFP:307-311 */
        }
    return best;
}
