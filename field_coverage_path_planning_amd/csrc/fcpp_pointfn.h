// fcpp_pointfn.h -- per-point device functions shared by the single-pass kernels (fcpp_fused.hip: eight points per lane, dense
// sampling; fcpp_sparse.hip: one point per lane, sparse sampling): template-based point evaluation, curvature from chords, the
// curvature clamp, wave reductions and the LDS-staged obstacle test.  Same arithmetic in both kernels => same results.
#pragma once
#include "fcpp_devfn.h"

namespace fcpp {

// nominal speeds by primitive kind (fs & FCPP_KIND_MASK): a per-lane LDS lookup instead of a divergent switch over scalars
struct NomTable { double v[8], ms[8]; };
// a wave-uniform value as an opaque scalar register: the compiler keeps (or spills to a vector-register lane) what it cannot re-derive
// (values that went through the vector unit -- f64 arithmetic -- come back through readfirstlane)
__device__ __forceinline__ double uniform_value(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int uniform_value(int v) { return __builtin_amdgcn_readfirstlane(v); }
#define FCPP_PIN(x) do { x = uniform_value(x); asm volatile("" : "+s"(x)); } while (0)
#define FCPP_PIN_V(x) asm volatile("" : "+v"(x))       // the same, kept in a vector register
static constexpr int TMPL_LDS = 64;        // turn-template samples a wavefront stages in LDS (k_plan_quiet; the reference's turn has 20)
__device__ __forceinline__ double nom_v(const NomTable &t, uint32_t fs) { return t.v[fs & FCPP_KIND_MASK]; }
__device__ __forceinline__ double nom_ms(const NomTable &t, uint32_t fs) { return t.ms[fs & FCPP_KIND_MASK]; }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// curvature (MLP:513-536) from the two chords and their lengths.  dtheta = atan2(sin(t2-t1), cos(t2-t1)) is the
// signed angle between the chords = atan2(cross, dot).  Exactly collinear chords (every interior point of an
// axis-aligned straight run) give 0; turning angles up to 0.124 rad (consecutive samples of any turn at fine sampling)
// take the odd series atan(x) = x - x^3/3 + x^5/5 - ..., 10 terms of which leave < 3e-20 at x = 1/8 (for |x| < 1e-8 the
// series returns x itself, as atan2 does).  Larger angles set `slow`: the caller evaluates atan2(cross, dot).
__device__ __forceinline__ double curv_chords_fast(double dx1, double dy1, double ds1, double dx2, double dy2, double ds2, bool &slow)
{
    slow = false;
    if (ds1 < 1e-6 || ds2 < 1e-6) return 0.0;
    const double cr = dx1 * dy2 - dy1 * dx2, dt = dx1 * dx2 + dy1 * dy2;
    if (cr == 0.0 && dt > 0.0) return 0.0;
    if (!(dt > 0.0 && fabs(cr) <= 0.125 * dt)) { slow = true; return 0.0; }
    const double x = cr / dt, z = x * x;
    double p = -1.0 / 19.0;
    p = fma(p, z, 1.0 / 17.0); p = fma(p, z, -1.0 / 15.0); p = fma(p, z, 1.0 / 13.0); p = fma(p, z, -1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0); p = fma(p, z, -1.0 / 7.0); p = fma(p, z, 1.0 / 5.0); p = fma(p, z, -1.0 / 3.0);
    const double dth = fma(x * z, p, x);
    return fabs(2 * dth / (ds1 + ds2));
}

// the same with the atan2 fallback as an out-of-line call (halo recomputation of k_plan_fused: one call site, not eight)
__device__ __forceinline__ double curv_chords(double dx1, double dy1, double ds1, double dx2, double dy2, double ds2)
{
    bool slow;
    const double k = curv_chords_fast(dx1, dy1, ds1, dx2, dy2, ds2, slow);
    if (!slow) return k;
    return fabs(2 * atan2_slow(dx1 * dy2 - dy1 * dx2, dx1 * dx2 + dy1 * dy2) / (ds1 + ds2));
}

// speed after the curvature clamp (MLP:496-504); nominal = the primitive's nominal speed.
// v_nom > sqrt(a_lat / kappa) * sf * 3.6  <=>  kappa * (v_nom / (sf * 3.6))^2 > a_lat: points that are clearly not clamped
// (the vast majority of turn points) are decided by that product, without the square root and the division.
__device__ __forceinline__ double clamped_speed(double v_nom, double kappa, const DevConst &cst, bool &clamped)
{
    clamped = false;
    if (kappa > 1e-6) {
        const double q = v_nom * cst.inv_sf36;
        if (kappa * q * q < cst.a_lat * (1.0 - 1e-9)) return v_nom;
        const double vmax_ms = sqrt(cst.a_lat / kappa) * cst.sf;
        const double vmax_kmh = vmax_ms * 3.6;
        if (v_nom > vmax_kmh) { clamped = true; return vmax_kmh; }
    }
    return v_nom;
}

// sqrt(fl(t*t)) == |t| exactly in IEEE arithmetic: axis-aligned steps need no square root
__device__ __forceinline__ double seg_len(double dx, double dy)
{
    return (dy == 0.0) ? fabs(dx) : ((dx == 0.0) ? fabs(dy) : sqrt(dx * dx + dy * dy));
}

// ---- template-based point evaluation ---------------------------------------------------------------------
__device__ __forceinline__ void rotate_back(const DevField &f, double &px, double &py)   // MLP:271-282, angle = +rotation
{
    const double tx = px - f.rot_cx, ty = py - f.rot_cy;
    px = (tx * f.rot_cos - ty * f.rot_sin) + f.rot_cx;
    py = (tx * f.rot_sin + ty * f.rot_cos) + f.rot_cy;
}

// layer 1 (MLP:750-780): pass position idx, offset off inside the pass
__device__ __forceinline__ void eval_main(const DevField &f, const DevConst &cst, int idx, int off, double &px, double &py,
                                          uint32_t &fw)
{
    const int pi = f.reverse_order ? (f.P - 1 - idx) : idx;
    const double y = f.min_y + (double)pi * f.W;
    const bool go_left = f.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
    if (off < f.n_line) {
        px = go_left ? linspace_at32(f.lex, f.lsx, -f.line_step, f.n_line, off) : linspace_at32(f.lsx, f.lex, f.line_step, f.n_line, off);
        py = y;
        fw = FCPP_KIND_SWATH | ((uint32_t)pi << FCPP_INDEX_SHIFT);
    } else {
        const double2 t = cst.tmpl_u[off - f.n_line];
        const bool turn_right = !go_left;                              // MLP:776
        if (f.turn_model == FCPP_TURN_ARC) px = turn_right ? (f.max_x - t.x) : (f.min_x + t.x);   // MLP:815, 822
        else px = turn_right ? ((f.max_x - f.R) + t.x) : ((f.min_x + f.R) - t.x);
        py = y + t.y;
        fw = FCPP_KIND_UTURN | ((uint32_t)pi << FCPP_INDEX_SHIFT);
    }
    if (f.rotated) rotate_back(f, px, py);
}

// layer 2 (MLP:943-1084): sample r of primitive p.  tc: the corner template's sample r if the caller has fetched it already
// (k_plan_sparse asks for it together with the primitive record instead of after it), else NULL
__device__ __forceinline__ void eval_prim(const DevPrim &p, const DevConst &cst, int r, double &px, double &py, const double2 *tc = nullptr)
{
    if (p.kind == PRIM_LINSPACE) {
        px = linspace_at32(p.a[0], p.a[2], p.a[4], p.n, r);
        py = linspace_at32(p.a[1], p.a[3], p.a[5], p.n, r);
    } else if (p.kind == PRIM_POINT) { px = p.a[0]; py = p.a[1]; }
    else if (p.kind == PRIM_RAY) {
        const double t = linspace_at32(0.0, p.a[4], p.a[5], p.n, r);
        px = p.a[0] + t * p.a[2];
        py = p.a[1] + t * p.a[3];
    } else if (p.kind == PRIM_UTURN) {     // a U-turn of layer 1 as a primitive: template sample + translation (mirror), as in eval_main
        const double2 t = cst.tmpl_u[r];
        const bool turn_right = p.form & 1;
        if (!(p.form & 4)) px = turn_right ? (p.a[0] - t.x) : (p.a[0] + t.x);
        else px = turn_right ? (p.a[0] + t.x) : (p.a[0] - t.x);
        py = p.a[1] + t.y;
        if (p.form & 2) {
            const double tx = px - p.a[4], ty = py - p.a[5];
            px = (tx * p.a[2] - ty * p.a[3]) + p.a[4];
            py = (tx * p.a[3] + ty * p.a[2]) + p.a[5];
        }
    } else {   // corner turn: quadrant formulas MLP:1049-1060 on the template (t1, t2) = (R(1-cos), R sin) or its clothoid analogue
        const double2 t = tc ? *tc : cst.tmpl_c[r];
        const int ci = p.kind == PRIM_ARC ? p.form : ((p.form + 3) & 3);
        if (ci == 0)      { px = p.a[0] + t.x; py = p.a[1] + t.y; }
        else if (ci == 1) { px = p.a[0] - t.y; py = p.a[1] + t.x; }
        else if (ci == 2) { px = p.a[0] - t.x; py = p.a[1] - t.y; }
        else              { px = p.a[0] + t.y; py = p.a[1] - t.x; }
    }
}

// The same for the one-point-per-lane kernel (fcpp_sparse_fn.h), whose wavefronts hold several primitive kinds side by side and pay for
// every branch that any lane takes: the three straight kinds -- point, numpy.linspace segment, reverse ray -- go through ONE form
//     t = r * ts ;  p = a0 + t * d        (linspace: ts = 1, d = step;  ray: ts = t-step, d = unit direction;  point: r = 0)
// with the last sample of a segment set to its end point (numpy.linspace stores `stop` itself) and the last ray parameter to the ray's
// length: the same roundings as linspace_at32 / eval_prim, sample for sample.  A segment whose step underflowed to zero although its
// ends differ (form bit 3, set by the host; no real field has one) takes eval_prim.
__device__ __forceinline__ void eval_prim_lanes(const DevPrim &p, const DevConst &cst, int r, double &px, double &py, const double2 &tc)
{
    const int kind = p.kind;
    if (kind == PRIM_POINT || kind == PRIM_LINSPACE || kind == PRIM_RAY) {
        if (__builtin_expect(p.form & 8, 0)) { eval_prim(p, cst, r, px, py, &tc); return; }
        const bool ray = kind == PRIM_RAY;
        const bool last = p.n > 1 && r == p.n - 1;
        const double ts = ray ? p.a[5] : 1.0, dx = ray ? p.a[2] : p.a[4], dy = ray ? p.a[3] : p.a[5];
        const double t = (ray && last) ? p.a[4] : (double)r * ts;
        px = p.a[0] + t * dx;
        py = p.a[1] + t * dy;
        if (last && !ray) { px = p.a[2]; py = p.a[3]; }
    } else eval_prim(p, cst, r, px, py, &tc);
}

// the batch's primitive table, or a field's primitives staged in LDS: indexed by the batch-wide primitive index either way
struct PrimTable {
    const DevPrim *p;
    int first;
    __device__ __forceinline__ const DevPrim &operator[](int i) const { return p[i - first]; }
};

__device__ __forceinline__ int find_prim(const DevField &f, const PrimTable prims, int64_t i)
{
    int a = f.prim_first, b = f.prim_first + f.prim_count - 1;
    while (a < b) {
        const int m = (a + b + 1) >> 1;
        if (prims[m].start <= i) a = m; else b = m - 1;
    }
    return a;
}

// ---- data-parallel-primitive moves: a lane reads a neighbour's register without a trip through the LDS crossbar (ds_bpermute).
// ctrl: 0x138 wave_shr:1 (lane i <- i-1), 0x130 wave_shl:1 (lane i <- i+1), quad_perm 0xB1 = [1,0,3,2], 0x4E = [2,3,0,1],
// 0x141 row_half_mirror, 0x140 row_mirror, 0x142 row_bcast:15, 0x143 row_bcast:31.  bound_ctrl is set: a lane without a source
// receives 0 and no `old` operand has to be materialised (without it every move costs a second v_mov for the destination's
// previous contents: the moves were 35 % of k_plan_sparse's vector instructions).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_prev(double v) { return dpp_mov<0x138>(v); }   // lane 0 receives 0
__device__ __forceinline__ double lane_next(double v) { return dpp_mov<0x130>(v); }   // lane 63 receives 0

// fmin / fmax on values known not to be signalling NaNs: the library calls put a canonicalising v_max_f64 x, x, x in front of every
// operand that did not come straight out of an arithmetic instruction (lane moves, selects), doubling the cost of a butterfly
__device__ __forceinline__ double max_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// wave totals by DPP butterflies (quad, half row, row) and two row broadcasts; the result is valid in LANE 63 only (the
// broadcasts go to every row, rows 0-2 end up with sums nobody reads; lane 63 adds ((r3 + r2) + (r1 + r0))).  Fixed order of
// the additions => run-to-run identical sums.  Values of the max variant must be >= 0.
__device__ __forceinline__ double wave_sum_to63(double v)
{
    if (__ballot(v != 0.0) == 0ull) return 0.0;
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    v += dpp_mov<0x142>(v);
    v += dpp_mov<0x143>(v);
    return v;
}
__device__ __forceinline__ double wave_max0_to63(double v)
{
    if (__ballot(v != 0.0) == 0ull) return 0.0;
    v = max_raw(v, dpp_mov<0xB1>(v));
    v = max_raw(v, dpp_mov<0x4E>(v));
    v = max_raw(v, dpp_mov<0x141>(v));
    v = max_raw(v, dpp_mov<0x140>(v));
    v = max_raw(v, dpp_mov<0x142>(v));
    v = max_raw(v, dpp_mov<0x143>(v));
    return v;
}

// Three (four) per-lane values reduced over the wave together: after the exchange with lane ^ 1 every lane carries two of the four
// slots (for its pair of lanes), after lane ^ 2 one slot (for its quad) -- the first two butterfly stages cost 7 + 7 additions and
// moves for all four values instead of 4 x 6 -- then the quads are added down the row (row_shr:4, row_shr:8) and the rows down the
// wave (two LDS-crossbar moves, which keep the lane's position in its row).  Lane 60 + j ends up with the total of slot
// WAVE4_SLOT(j) = 0, 2, 1, 3 for j = 0..3; the other lanes hold partial results nobody reads.  Fixed order => identical sums run
// to run.  OP: 0 add, 1 max of non-negative values.
#define WAVE4_SLOT(lane) ((((lane) & 1) << 1) | (((lane) >> 1) & 1))
template <int OP>
__device__ __forceinline__ double wave4_to_hi(double s0, double s1, double s2, double s3)
{
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    auto op = [](double a, double b) { return OP == 0 ? a + b : max_raw(a, b); };
    const double a = op(b0 ? s2 : s0, dpp_mov<0xB1>(b0 ? s0 : s2));      // slot 0 (even lanes) / slot 2 (odd lanes) of the lane pair
    const double c = op(b0 ? s3 : s1, dpp_mov<0xB1>(b0 ? s1 : s3));      // slot 1 / slot 3
    double v = op(b1 ? c : a, dpp_mov<0x4E>(b1 ? a : c));                  // one slot per lane, for its quad
    v = op(v, dpp_mov<0x114>(v));                                          // row_shr:4
    v = op(v, dpp_mov<0x118>(v));                                          // row_shr:8: lanes 12-15 of a row hold the row's four totals
    const double u16 = __shfl_up(v, 16);
    if (lane >= 16) v = op(v, u16);
    const double u32 = __shfl_up(v, 32);
    if (lane >= 32) v = op(v, u32);
    return v;
}

// a / b for a divisor whose correctly rounded reciprocal rb = RN(1 / b) is at hand (constants, per-batch speeds): q = RN(a rb) is
// within an ulp, the residual a - b q is exact in one fma, and RN(q + r rb) is the correctly rounded quotient (Markstein 1990;
// the exception, a significand of b that is all ones, does not occur among the divisors used).  3 instructions instead of the
// 12-15 of the IEEE division sequence, bit-identical results (tests/test_recip_div.py).
__device__ __forceinline__ double div_by(double a, double b, double rb)
{
    const double q = a * rb;
    return fma(fma(-q, b, a), rb, q);
}
__device__ __forceinline__ double div36(double a) { return div_by(a, 3.6, 1.0 / 3.6); }   // km/h -> m/s

// ... and with the fallback in line (one point per lane: fcpp_sparse_fn.h)
__device__ __forceinline__ double curv_chords_inline(double dx1, double dy1, double ds1, double dx2, double dy2, double ds2)
{
    bool slow;
    const double k = curv_chords_fast(dx1, dy1, ds1, dx2, dy2, ds2, slow);
    if (!slow) return k;
    return fabs(2 * atan2_fd(dx1 * dy2 - dy1 * dx2, dx1 * dx2 + dy1 * dy2) / (ds1 + ds2));
}

// a / b for operands of ordinary magnitude (lengths, speeds, their products: nothing near the ends of the exponent range), without the
// scaling and fix-up instructions of the IEEE sequence: the hardware reciprocal (< 1 ulp), one Newton step, then Markstein's correction
// of the quotient by its exact residual -- the result is the correctly rounded quotient except for rare last-bit cases, in 1 + 5
// instructions instead of 12 (k_plan_sparse executes about ten divisions per wavefront: curvatures, clamps, segment times).
__device__ __forceinline__ double fdiv(double a, double b)
{
    double x = __builtin_amdgcn_rcp(b);
    x = fma(fma(-b, x, 1.0), x, x);
    const double q = a * x;
    return fma(fma(-q, b, a), x, q);
}

// sqrt(x) for a positive x of ordinary magnitude (squared lengths and speeds: nothing below 2^-767, never 0 or infinite): the
// hardware's reciprocal square root, the two coupled Newton steps and the two residual corrections of the compiler's own IEEE sequence
// -- the same instructions, so the same correctly rounded result -- without its range scaling and special-value selects (10 instead of
// 20 instructions; k_plan_sparse takes up to six square roots per lane)
__device__ __forceinline__ double fsqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    return fma(fma(-g, g, x), h, g);
}
__device__ __forceinline__ double seg_len_fast(double dx, double dy)      // |(dx, dy)|: seg_len's value without its shortcuts
{
    // (seg_len returns |dx| where dy == 0 and |dy| where dx == 0 to save the square root; as selects in a wavefront that holds both
    // kinds of steps they only add instructions -- and sqrt(fl(t t)) == |t| exactly for a correctly rounded root, which fsqrt_pos is)
    const double s = dx * dx + dy * dy;
    return s == 0.0 ? 0.0 : fsqrt_pos(s);
}

// a * b + k, k a compile-time constant that the scalar unit puts into a scalar register pair (one scalar operand per vector instruction)
__device__ __forceinline__ double fma_sk(double a, double b, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

// atan2_fd's (fcpp_geom.h) five reduction intervals as a table of six doubles each: t = num / den with num = na a + nb b and
// den = da a + db b, then hi, lo.  The products by 0, 1, 2 are exact and 1.5 x rounds as in atan2_fd, so num and den are the values
// its expressions give; one lookup by the interval's number replaces four selects each of num, den, hi and lo.
static constexpr int ATAN_TAB_DOUBLES = 32;
static __constant__ double k_atan_tab[ATAN_TAB_DOUBLES] = {
    1.0,  0.0, 0.0, 1.0, 0.0,                          0.0,
    2.0, -1.0, 1.0, 2.0, 4.63647609000806093515e-01,   2.26987774529616870924e-17,
    1.0, -1.0, 1.0, 1.0, 7.85398163397448278999e-01,   3.06161699786838301793e-17,
    1.0, -1.5, 1.5, 1.0, 9.82793723247329054082e-01,   1.39033110312309984516e-17,
    0.0, -1.0, 1.0, 0.0, 1.57079632679489655800e+00,   6.12323399573676603587e-17,
    0.0,  0.0 };
// every wavefront writes the whole table itself (all write the same values): no barrier between the workgroup's wavefronts
__device__ __forceinline__ void atan_tab_stage(double *tab /* LDS, ATAN_TAB_DOUBLES */)
{
    const int lane = threadIdx.x & 63;
    if (lane < ATAN_TAB_DOUBLES) tab[lane] = k_atan_tab[lane];
}

// |atan2_fd(y, x)| (fcpp_geom.h) with fdiv for its one quotient and the table for its reduction: same intervals, same polynomial
__device__ __forceinline__ double atan2_abs_dev(double y, double x, const double *tab)
{
    const double a = fabs(y), b = fabs(x);
    const int id = (int)!(a < 0.4375 * b) + (int)!(a < 0.6875 * b) + (int)!(a < 1.1875 * b) + (int)!(a < 2.4375 * b);
    const double2 n = *reinterpret_cast<const double2 *>(tab + 6 * id), d = *reinterpret_cast<const double2 *>(tab + 6 * id + 2),
                  hl = *reinterpret_cast<const double2 *>(tab + 6 * id + 4);
    const double num = n.x * a + n.y * b, den = d.x * a + d.y * b;
    const double t = fdiv(num, den), z = t * t, w = z * z;
    // (Horner steps with the coefficient as the addend from SCALAR registers: the compiler's v_fmac form wants it in the destination, i.e.
    // two vector moves per 64-bit literal -- 18 of the 95 instructions of one curvature)
    const double s1 = z * fma_sk(w, fma_sk(w, fma_sk(w, fma_sk(w, fma_sk(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02), 6.66107313738753120669e-02),
                                                          9.09088713343650656196e-02), 1.42857142725034663711e-01), 3.33333333333329318027e-01);
    const double s2 = w * fma_sk(w, fma_sk(w, fma_sk(w, fma_sk(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02), -7.69187620504482999495e-02),
                                                -1.11111104054623557880e-01), -1.99999999998764832476e-01);
    double r = hl.x - ((t * (s1 + s2) - hl.y) - t);
    if (x < 0.0) r = 3.14159265358979311600e+00 - (r - 1.22464679914735317720e-16);
    return r;
}

// the curvature clamp of clamped_speed with fdiv and fsqrt_pos (k_plan_sparse)
__device__ __forceinline__ double clamped_speed_fast(double v_nom, double kappa, const DevConst &cst, bool &clamped)
{
    clamped = false;
    if (kappa > 1e-6) {
        const double q = v_nom * cst.inv_sf36;
        if (kappa * q * q < cst.a_lat * (1.0 - 1e-9)) return v_nom;
        const double vmax_ms = fsqrt_pos(fdiv(cst.a_lat, kappa)) * cst.sf;
        const double vmax_kmh = vmax_ms * 3.6;
        if (v_nom > vmax_kmh) { clamped = true; return vmax_kmh; }
    }
    return v_nom;
}

// the curvature of curv_chords through atan2_fd alone (k_plan_sparse).  At the reference's sampling every wavefront of that kernel
// holds junctions between primitives, whose turning angles are far beyond the short series of curv_chords_fast, so it paid for the
// series AND for the atan2 fallback; atan2_fd's first interval (|cross| < 7/16 dot) is a series of the same length anyway.  Exactly
// collinear chords still give exactly 0 (atan2_fd(0, dot > 0) = 0).  |2 atan2 / s| = 2 |atan2| / s: every step of fdiv is odd in its
// numerator, so the sign of the angle is never formed.
__device__ __forceinline__ double curv_chords_atan(double dx1, double dy1, double ds1, double dx2, double dy2, double ds2, const double *atab)
{
    if (ds1 < 1e-6 || ds2 < 1e-6) return 0.0;
    const double cr = dx1 * dy2 - dy1 * dx2, dt = dx1 * dx2 + dy1 * dy2;
    if (cr == 0.0 && dt > 0.0) return 0.0;
    return fdiv(2 * atan2_abs_dev(cr, dt, atab), ds1 + ds2);
}

// wave-wide reductions; a ballot skips the butterfly when every lane holds the neutral element (most tiles have
// only one layer, no curvature and no flags)
__device__ __forceinline__ double wave_sum(double v)
{
    if (__ballot(v != 0.0) == 0ull) return 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max0(double v)   // v >= 0
{
    if (__ballot(v != 0.0) == 0ull) return 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ long long wave_sum_i(int v)
{
    if (__ballot(v != 0) == 0ull) return 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- obstacle flags (build-defined validator): even-odd crossing test of the wave's points against the field's
// obstacle polygons.  Culling first: lane b compares polygon b's bounding box with the bounding box of the wave's
// points, a ballot collects the few candidates, and each candidate's vertices are staged in LDS (coalesced load, then
// broadcast reads) before every lane tests its own points against all its edges.
static constexpr int OBS_LDS_VERTS = 240;    // polygons with more vertices are read from global memory (240: four wavefronts' staging areas and the rest of k_plan_sparse_fields' LDS fit 20 KiB, eight workgroups per CU)

// returns the bit mask of the points (bit p = point p of this lane) that lie inside an obstacle
template <int NP>
__device__ __forceinline__ unsigned obstacle_mask(const DevObstacles &obs, int ob0, int ob1, double *lds /* 2*OBS_LDS_VERTS */,
                                                  double bminx, double bminy, double bmaxx, double bmaxy,
                                                  const double (&px)[NP], const double (&py)[NP], int nvalid)
{
    const int lane = threadIdx.x & 63;
    unsigned inside = 0;    // bit k: point k is inside some obstacle
    for (int base = ob0; base < ob1; base += 64) {
        const int b = base + lane;
        bool hit = false;
        if (b < ob1) {
            const double *bb = obs.bbox + 4 * (int64_t)b;
            hit = !(bb[0] > bmaxx || bb[2] < bminx || bb[1] > bmaxy || bb[3] < bminy);
        }
        unsigned long long cand = __ballot(hit);
        while (cand) {
            const int k = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            // which of this wave's points lie in the polygon's own box (with a margin far above any rounding of the crossing test): a
            // tile's box can be huge -- the tile that holds the jump from the last swath to the headland spans the field, every one of
            // cfg3's 32 polygons passed the cull above and its 163 points cost 340 us of crossing tests -- the points' boxes are not
            const double *pb = obs.bbox + 4 * (int64_t)(base + k);
            const double qx0 = pb[0] - 1e-6, qy0 = pb[1] - 1e-6, qx1 = pb[2] + 1e-6, qy1 = pb[3] + 1e-6;
            unsigned near = 0;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (p < nvalid && px[p] >= qx0 && px[p] <= qx1 && py[p] >= qy0 && py[p] <= qy1) near |= 1u << p;
            if (__ballot(near != 0u) == 0ull) continue;
            const int64_t a0 = obs.offsets[base + k], a1 = obs.offsets[base + k + 1];
            const int nv = (int)(a1 - a0);
            const bool staged = nv <= OBS_LDS_VERTS;
            if (staged) {
                wave_sync();
                for (int q = lane; q < nv; q += 64) { lds[2 * q] = obs.x[a0 + q]; lds[2 * q + 1] = obs.y[a0 + q]; }
                wave_sync();
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const bool test = ((near >> p) & 1u) && !((inside >> p) & 1u);
                if (__ballot(test) == 0ull) continue;
                if (test) {
                    bool in = false;
                    for (int q = 0, r = nv - 1; q < nv; r = q++) {
                        const double xi = staged ? lds[2 * q] : obs.x[a0 + q], yi = staged ? lds[2 * q + 1] : obs.y[a0 + q];
                        const double xj = staged ? lds[2 * r] : obs.x[a0 + r], yj = staged ? lds[2 * r + 1] : obs.y[a0 + r];
                        if (((yi > py[p]) != (yj > py[p])) && (px[p] < (xj - xi) * (py[p] - yi) / (yj - yi) + xi)) in = !in;
                    }
                    if (in) inside |= 1u << p;
                }
            }
        }
    }
    return inside;
}

}  // namespace fcpp
