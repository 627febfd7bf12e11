// fcpp_devplan.hip -- batch setup on the device, see fcpp_devplan.h.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "fcpp_devplan.h"
#include "fcpp_quiet_fn.h"

namespace fcpp {

size_t devplan_scratch_layout(int64_t n, int max_prims, DevPlanScratch *o)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t r = off; off = (off + bytes + 255) & ~(size_t)255; return r; };
    const size_t nn = (size_t)(n > 0 ? n : 1), nblk = (nn + 1023) / 1024;
    o->totals = reinterpret_cast<int64_t *>(take(PLAN_TOTALS * sizeof(int64_t)));      // (first: the flags keep their place whatever n is)
    o->fields_in = reinterpret_cast<fcpp_field *>(take(nn * sizeof(fcpp_field)));
    o->info = reinterpret_cast<fcpp_field_info *>(take(nn * sizeof(fcpp_field_info)));
    o->fields_tmp = reinterpret_cast<DevField *>(take(nn * sizeof(DevField)));
    o->prims_tmp = reinterpret_cast<DevPrim *>(take(nn * (size_t)max_prims * sizeof(DevPrim)));
    o->counts = reinterpret_cast<int64_t *>(take(nn * PC_COLS * sizeof(int64_t)));
    o->bases = reinterpret_cast<int64_t *>(take(nn * PC_COLS * sizeof(int64_t)));
    o->blk_sums = reinterpret_cast<int64_t *>(take(nblk * PC_COLS * sizeof(int64_t)));
    o->keep_tiles = reinterpret_cast<DevTile *>(take(nn * DEVPLAN_KEEP_ROWS * sizeof(DevTile)));
    o->keep_wtiles = reinterpret_cast<DevWaveTile *>(take(nn * DEVPLAN_KEEP_WROWS * sizeof(DevWaveTile)));
    return off;
}

// blocks of 1024 fields up to which a batch is "small": ONE scan launch (a workgroup per column walks the fields in chunks of 4096) and
// -- fcpp_api.cpp -- the speculative capacity layout.  FCPP_SMALL_BLOCKS (read once) for the A/B.
int64_t devplan_small_blocks()
{
    static const int64_t v = [] { const char *e = getenv("FCPP_SMALL_BLOCKS"); const int64_t x = e ? atoll(e) : 8; return x < 1 ? 1 : (x > 128 ? 128 : x); }();
    return v;
}

namespace {

// ---- k_plan_fields: one thread per field, the host's own plan function ----------------------------------------------------------------
struct DevSink {
    DevPrim *base;
    int cap;
    int64_t n;
    __device__ int64_t size() const { return n; }
    __device__ void push(const DevPrim &p) { if (n < cap) base[n] = p; ++n; }
    __device__ void truncate(int64_t m) { n = m; }
    __device__ int clipped_layer1(const PlanConsts &, const fcpp_field &, const Layer1Frame &, int64_t &) { return FCPP_EUNSUPPORTED; }
    __device__ int headland_straight(const PlanConsts &, const Quad &, const DevPrim &, int64_t &) { return FCPP_EUNSUPPORTED; }
    __device__ bool box_meets_square(double, double, double) const { return false; }
    __device__ bool box_meets_segment(double, double, double, double) const { return false; }
};

__global__ __launch_bounds__(64) void k_plan_fields(int64_t n, PlanConsts pc, const fcpp_field *__restrict__ fin, fcpp_field_info *__restrict__ info,
                                                    DevField *__restrict__ ftmp, DevPrim *__restrict__ ptmp, int64_t *__restrict__ counts,
                                                    int64_t *__restrict__ totals, int64_t n_polys, int check_obstacles, int count_only, int64_t gen)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const fcpp_field f = fin[i];
    // the field's obstacle range must lie inside the batch's polygon table: the kernels of a step index it
    if (check_obstacles && (f.n_obstacles < 0 || f.obstacle_first < 0 || (f.n_obstacles > 0 && f.obstacle_first + f.n_obstacles > n_polys)))
        atomicMax(reinterpret_cast<unsigned long long *>(totals + PC_COLS + PF_BAD_OBSTACLES), (unsigned long long)gen);
    DevSink sink{ count_only ? nullptr : ptmp + i * pc.max_prims, count_only ? 0 : pc.max_prims, 0 };
    const int64_t npts = plan_field_t(pc, f, info[i], ftmp[i], sink);
    counts[(int64_t)PC_POINTS * n + i] = npts;
    if (count_only) return;
    counts[(int64_t)PC_PRIMS * n + i] = ftmp[i].prim_count;
    if (sink.n > pc.max_prims) atomicMax(reinterpret_cast<unsigned long long *>(totals + PC_COLS + PF_FALLBACK), (unsigned long long)gen);
}

// ---- k_plan_fields16: the same plan, SIXTEEN lanes per field (round 5).  plan_field_t is a chain of some 5000 dependent instructions
// per field -- four corner angles, four mitre directions, an inset per headland loop, eight primitives per loop -- and one thread per
// field walks it alone (35 us for any batch up to 64 x 1024 fields: 64 wavefronts on 1024 SIMDs).  Here a field is a ROW of sixteen
// lanes = four quads: lane i of a quad owns vertex / edge / side i of the quadrilateral, quad l the headland loop l (four loops at a
// time), neighbours come by lane moves inside the quad.  Every value is computed by the same float64 operations in the same order as in
// plan_field_t (sums over the four vertices run 0, 1, 2, 3 on gathered values): the records are equal byte for byte to the host's
// (tests/test_gpu_devplan.py), and FCPP_PLAN_SERIAL=1 keeps the one-thread kernel as the A/B.
#ifdef FCPP_DIAG_TILE
__device__ unsigned long long g_plan_stamps[16];
#define PSTAMP(k) do { if (field == 1000 && (threadIdx.x & 15) == 0) g_plan_stamps[k] = wall_clock64(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
namespace p16 {
__device__ __forceinline__ double qget(double v, int k) { return __shfl(v, (int)((threadIdx.x & ~3u) | (unsigned)k)); }          // lane k of this lane's quad
__device__ __forceinline__ int qgeti(int v, int k) { return __shfl(v, (int)((threadIdx.x & ~3u) | (unsigned)k)); }
__device__ __forceinline__ double rget(double v, int k) { return __shfl(v, (int)((threadIdx.x & ~15u) | (unsigned)k)); }         // lane k of this lane's row
__device__ __forceinline__ int rgeti(int v, int k) { return __shfl(v, (int)((threadIdx.x & ~15u) | (unsigned)k)); }
__device__ __forceinline__ unsigned qballot(bool p) { return (unsigned)((__ballot(p) >> (threadIdx.x & 60u)) & 0xFull); }     // this quad's four bits
__device__ __forceinline__ unsigned rballot(bool p) { return (unsigned)((__ballot(p) >> (threadIdx.x & 48u)) & 0xFFFFull); }  // this row's sixteen bits
// s = 0; s += v[0]; ... s += v[3] over the quad's four values, as the host's loops add them
__device__ __forceinline__ double qsum_seq(double v) { double s = 0.0; s += qget(v, 0); s += qget(v, 1); s += qget(v, 2); s += qget(v, 3); return s; }
__device__ __forceinline__ double qmin_seq(double v) { double m = qget(v, 0); for (int k = 1; k < 4; ++k) m = hmin(m, qget(v, k)); return m; }
__device__ __forceinline__ double qmax_seq(double v) { double m = qget(v, 0); for (int k = 1; k < 4; ++k) m = hmax(m, qget(v, k)); return m; }
// area_centroid of the quad's polygon (px, py = this lane's vertex)
__device__ __forceinline__ double area_centroid_q(double px, double py, int i, double &cx, double &cy)
{
    const double nx = qget(px, (i + 1) & 3), ny = qget(py, (i + 1) & 3);
    const double cr = px * ny - nx * py;
    double a = qsum_seq(cr);
    const double sx = qsum_seq((px + nx) * cr), sy = qsum_seq((py + ny) * cr);
    a *= 0.5;
    if (fabs(a) < 1e-300) { cx = qget(px, 0); cy = qget(py, 0); return 0.0; }
    cx = sx / (6.0 * a); cy = sy / (6.0 * a);
    return a;
}
}  // namespace p16

// (f0, f1: the fields [f0, f1) of the batch's n -- large batches are planned in chunks, launch_devplan_count)
__global__ __launch_bounds__(64) void k_plan_fields16(int64_t n, PlanConsts pc, const fcpp_field *__restrict__ fin, fcpp_field_info *__restrict__ info,
                                                      DevField *__restrict__ ftmp, DevPrim *__restrict__ ptmp, int64_t *__restrict__ counts,
                                                      int64_t *__restrict__ totals, int64_t n_polys, int check_obstacles, int64_t gen, int64_t f0, int64_t f1)
{
    using namespace p16;
    const int lane = threadIdx.x, l16 = lane & 15, i = lane & 3, lp = l16 >> 2;
    const int64_t field = f0 + (int64_t)blockIdx.x * 4 + (lane >> 4);
    if (field >= f1) return;
    PSTAMP(0);
    const fcpp_field f = fin[field];
    if (check_obstacles && l16 == 0 && (f.n_obstacles < 0 || f.obstacle_first < 0 || (f.n_obstacles > 0 && f.obstacle_first + f.n_obstacles > n_polys)))
        atomicMax(reinterpret_cast<unsigned long long *>(totals + PC_COLS + PF_BAD_OBSTACLES), (unsigned long long)gen);
    const fcpp_vehicle &veh = pc.veh;
    const fcpp_options &opt = pc.opt;
    const double W = pc.W, R = pc.R, ds = pc.ds;
    const bool cloth = pc.cloth != 0;
    // The two records are written where they go, by the row's lane 0, as plan_field_t writes them through its references: zeroed first
    // (all sixteen lanes, a few words each), then member by member in the function's order -- a field that raises keeps what had been
    // written before it did.  (Held in registers until the end they were 130 of them per lane: the kernel ran out, 50 us instead of 35.)
    fcpp_field_info &in = info[field];
    DevField &df = ftmp[field];
    {
        static_assert(sizeof(fcpp_field_info) % 8 == 0 && sizeof(DevField) % 8 == 0, "zeroed as 8-byte words");
        unsigned long long *zi = reinterpret_cast<unsigned long long *>(&in), *zd = reinterpret_cast<unsigned long long *>(&df);
        for (int w = l16; w < (int)(sizeof(fcpp_field_info) / 8); w += 16) zi[w] = 0ull;
        for (int w = l16; w < (int)(sizeof(DevField) / 8); w += 16) zd[w] = 0ull;
    }
    const bool w0 = l16 == 0;          // the lane that writes the records
    DevPrim *const pbase = ptmp + field * pc.max_prims;
    // the function's exit
    auto finish = [&](int64_t npts, int32_t prim_count, int32_t n_pushed) {
        if (w0) {
            counts[(int64_t)PC_POINTS * n + field] = npts;
            counts[(int64_t)PC_PRIMS * n + field] = prim_count;
            if (n_pushed > pc.max_prims) atomicMax(reinterpret_cast<unsigned long long *>(totals + PC_COLS + PF_FALLBACK), (unsigned long long)gen);
        }
    };
#define P16_FAIL(code) do { if (w0) { in.status = (code); in.n_main = in.n_head = 0; in.n_reverse[0] = in.n_reverse[1] = in.n_reverse[2] = in.n_reverse[3] = 0; \
                                      df.n_main = df.n_total = 0; df.gen_main = 0; df.prim_first = 0; df.prim_count = 0; } finish(0, 0, 0); return; } while (0)
    // this lane's vertex i and its neighbours
    const double qx = i == 0 ? f.vx[0] : (i == 1 ? f.vx[1] : (i == 2 ? f.vx[2] : f.vx[3]));
    const double qy = i == 0 ? f.vy[0] : (i == 1 ? f.vy[1] : (i == 2 ? f.vy[2] : f.vy[3]));
    const int in1 = (i + 1) & 3, ip1 = (i + 3) & 3, io = (i + 2) & 3;
    const double xn = qget(qx, in1), yn = qget(qy, in1), xp = qget(qx, ip1), yp = qget(qy, ip1), xo = qget(qx, io), yo = qget(qy, io);
    {
        const bool fin_i = isfinite(qx) && isfinite(qy);
        // is_convex: cr_i over (i, j = i + 1, k = i + 2)
        const double cr = (xn - qx) * (yo - yn) - (yn - qy) * (xo - xn);
        const int pos = __popc(qballot(cr > 0)), neg = __popc(qballot(cr < 0));
        const bool convex = (pos == 0 || neg == 0) && (pos + neg) > 0;
        if (qballot(fin_i) != 0xFu || !convex) P16_FAIL(FCPP_EUNSUPPORTED);
    }
    PSTAMP(1);
    // ---- __init__ (MLP:109-135, 137-163, 310, 322-343)
    const double bminx = qmin_seq(qx), bmaxx = qmax_seq(qx), bminy = qmin_seq(qy), bmaxy = qmax_seq(qy);
    const double L = f.from_vertices ? (bmaxx - bminx) : f.vx[1];
    const double H = f.from_vertices ? (bmaxy - bminy) : f.vy[2];
    if (w0) { in.field_length = L; in.field_width = H; }
    double cang;
    {   // corner_angle(q, i)
        const double v1x = xp - qx, v1y = yp - qy, v2x = xn - qx, v2y = yn - qy;
        double c = (v1x * v2x + v1y * v2y) / (sqrt(v1x * v1x + v1y * v1y) * sqrt(v2x * v2x + v2y * v2y));
        c = hmin(1.0, hmax(-1.0, c));
        cang = fc_acos(c) * (180.0 / kPi);
    }
    const double cang0 = qget(cang, 0), cang1 = qget(cang, 1), cang2 = qget(cang, 2), cang3 = qget(cang, 3);
    if (w0) { in.corner_angles[0] = cang0; in.corner_angles[1] = cang1; in.corner_angles[2] = cang2; in.corner_angles[3] = cang3; }
    const bool all90 = qballot(fabs(cang - 90) < 1.0) == 0xFu;
    bool is_par;
    {   // is_parallelogram: edge k against edge k + 2, k = 0, 1 (this lane: its own edge against the opposite one)
        const double ex = xn - qx, ey = yn - qy, ox = xp - xo, oy = yp - yo;
        const double cross = fabs(ex * oy - ey * ox);
        const double na = sqrt(ex * ex + ey * ey), nb = sqrt(ox * ox + oy * oy);
        const bool ok = cross < 0.01 * (na * nb);
        is_par = (qballot(ok) & 3u) == 3u;
    }
    if (w0) { in.shape = all90 ? 0 : (is_par ? 1 : 2); }
    const double hw = R;
    if (w0) { in.headland_width = hw; }
    const bool has_start = f.has_start && (0 <= f.start_x && f.start_x <= L && 0 <= f.start_y && f.start_y <= H);
    const bool has_end = f.has_end && (0 <= f.end_x && f.end_x <= L && 0 <= f.end_y && f.end_y <= H);
    if (w0) { in.start_kept = has_start; in.end_kept = has_end; }
    // ---- start corner (MLP:345-385)
    int sci = 0;
    if (has_start) {
        const double cxs = (i == 0 || i == 3) ? hw / 2 : L - hw / 2, cys = i < 2 ? hw / 2 : H - hw / 2;
        const double dx = cxs - f.start_x, dy = cys - f.start_y;
        const double d = sqrt(dx * dx + dy * dy);
        double best = 0;
        for (int k = 0; k < 4; ++k) { const double dk = qget(d, k); if (k == 0 || dk < best) { best = dk; sci = k; } }
    }
    if (w0) { in.start_corner = sci; }
    PSTAMP(2);
    // ---- layer 1 frame (MLP:591-611, 670-718): mitre_of(q), inset by hw
    double acx, acy;
    const double sgn = area_centroid_q(qx, qy, i, acx, acy) > 0 ? 1.0 : -1.0;
    double m_sx, m_sy, m_den;
    {
        const double ex = xn - qx, ey = yn - qy;
        const double ln = fc_hypot(ex, ey);
        const double nx = -ey / ln * sgn, ny = ex / ln * sgn;
        const double nxp = qget(nx, ip1), nyp = qget(ny, ip1);
        m_den = 1.0 + (nxp * nx + nyp * ny);
        m_sx = nxp + nx; m_sy = nyp + ny;
    }
    // inset(q, mit, d): this lane's vertex; false = empty
    auto inset_q = [&](double d, double &ox, double &oy) -> bool {
        ox = qx + d * m_sx / m_den;
        oy = qy + d * m_sy / m_den;
        const double ex = xn - qx, ey = yn - qy;
        const double oxn = qget(ox, in1), oyn = qget(oy, in1);
        return qballot((oxn - ox) * ex + (oyn - oy) * ey <= 0) == 0u;
    };
    double mqx, mqy;
    {
        const bool ok = inset_q(hw, mqx, mqy);
        double cx, cy;
        const double ar = fabs(area_centroid_q(mqx, mqy, i, cx, cy));
        if (!ok || ar < 1.0) P16_FAIL(FCPP_EINVAL);
    }
    PSTAMP(3);
    const double e0x = f.vx[1] - f.vx[0], e0y = f.vy[1] - f.vy[0];
    const double rot = (e0x == 0.0 && e0y == 0.0) ? 0.0 : fc_atan2_cr(e0y, e0x);          // (correctly rounded: fcpp_math.h, round 5)
    if (w0) { in.rotation_angle = rot; }
    const bool rotated = fabs(rot) > 0.01;
    if (w0) { in.rotated = rotated; }
    double rc, rs;
    fc_sincos_cr(rot, rs, rc);
    double ccx = 0, ccy = 0, sx = f.start_x, sy = f.start_y;
    double rqx = mqx, rqy = mqy;
    if (rotated) {
        area_centroid_q(mqx, mqy, i, ccx, ccy);
        rotate_point(mqx, mqy, rc, -rs, ccx, ccy, rqx, rqy);
        if (has_start) rotate_point(sx, sy, rc, -rs, ccx, ccy, sx, sy);
    }
    const double min_x = qmin_seq(rqx), max_x = qmax_seq(rqx), min_y = qmin_seq(rqy), max_y = qmax_seq(rqy);
    int reverse_order = 0, start_from_right = 0;   // MLP:631-668
    if (has_start) {
        if (sy > (min_y + max_y) / 2) reverse_order = 1;
        if (sx > (min_x + max_x) / 2) start_from_right = 1;
    }
    if (w0) { in.reverse_order = reverse_order; in.start_from_right = start_from_right; }
    PSTAMP(4);
    // ---- layer 1 sizes (MLP:736-739)
    const double lsx = min_x + R, lex = max_x - R;
    const double Pd = (max_y - min_y) / W;
    const int64_t P = Pd < (double)INT32_MAX ? (int64_t)Pd + 1 : (int64_t)INT32_MAX + 1;
    const int64_t n_line = ds > 0 ? n_for_length(fabs(lex - lsx), ds) : 2;
    const int64_t n_turn = ds > 0 ? n_for_length(pc.len_uturn, ds) : 20;
    if (P >= ((int64_t)1 << (32 - FCPP_INDEX_SHIFT)) || n_line + n_turn > INT32_MAX - 2 * TILE_POINTS) P16_FAIL(FCPP_ESIZE);
    if (w0) { in.n_swaths = (int32_t)P; }
    const int64_t n_main = P * n_line + (P - 1) * n_turn;
    if (w0) { df.gen_main = n_main; }
    if (w0) { df.prim_first = 0; }
    if (w0) { in.n_main = n_main; }
    if (w0) { df.n_main = n_main; }
    if (w0) { df.lsx = lsx; df.lex = lex; df.line_step = lin_step(lsx, lex, n_line); }
    if (w0) { df.min_x = min_x; df.max_x = max_x; df.min_y = min_y; df.W = W; df.R = R; }
    if (w0) { df.turn_end = pc.turn_end_pi; }
    if (w0) { df.turn_step = lin_step(0.0, pc.turn_end_pi, n_turn); }
    if (w0) { df.turn_Re = pc.Re_pi; }
    if (w0) { df.rot_cos = rc; df.rot_sin = rs; df.rot_cx = ccx; df.rot_cy = ccy; }
    if (w0) { df.v_work = veh.max_work_speed_kmh; df.v_turn = veh.headland_turn_speed_kmh; }
    if (w0) { df.P = (int32_t)P; df.n_line = (int32_t)n_line; df.n_turn = (int32_t)n_turn; }
    if (w0) { df.reverse_order = reverse_order; df.start_from_right = start_from_right; df.rotated = rotated; }
    if (w0) { df.turn_model = opt.turn_model; }
    PSTAMP(5);
    // geofence half-planes: edge i
    double gex, gey, geo;
    {
        const double ex = xn - qx, ey = yn - qy;
        const double ln = sqrt(ex * ex + ey * ey);
        gex = -ey / ln * sgn; gey = ex / ln * sgn;
        geo = -(gex * qx + gey * qy);
    }
    const double gex0 = qget(gex, 0), gex1 = qget(gex, 1), gex2 = qget(gex, 2), gex3 = qget(gex, 3);
    const double gey0 = qget(gey, 0), gey1 = qget(gey, 1), gey2 = qget(gey, 2), gey3 = qget(gey, 3);
    const double geo0 = qget(geo, 0), geo1 = qget(geo, 1), geo2 = qget(geo, 2), geo3 = qget(geo, 3);
    if (w0) { df.ex[0] = gex0; df.ex[1] = gex1; df.ex[2] = gex2; df.ex[3] = gex3; }
    if (w0) { df.ey[0] = gey0; df.ey[1] = gey1; df.ey[2] = gey2; df.ey[3] = gey3; }
    if (w0) { df.eo[0] = geo0; df.eo[1] = geo1; df.eo[2] = geo2; df.eo[3] = geo3; }

    // ---- layer 2 (MLP:898-1084): quad lp = loop l0 + lp, lane i = side i of the loop (its straight, then the turn at its end)
    PSTAMP(6);
    const int num_loops = (int)ceil(hw / W);
    if (w0) { in.n_loops = num_loops; }
    int64_t pos = n_main;
    int32_t prim_pos = 0;
    double first_head0 = 0, first_head1 = 0, last_head0 = 0, last_head1 = 0;
    int32_t nrev0 = 0, nrev1 = 0, nrev2 = 0, nrev3 = 0;
    const int64_t nt = pc.nt_corner;
    for (int l0 = 0; l0 < num_loops; l0 += 4) {
        const int loop = l0 + lp;
        const bool act = loop < num_loops;
        const double offset = W / 2 + loop * W;
        double cx, cy;
        bool inset_ok = inset_q(offset, cx, cy);
        {
            double tx, ty;
            const double ar = fabs(area_centroid_q(cx, cy, i, tx, ty));
            if (ar < 1.0) inset_ok = false;
        }
        if (opt.ring_order == FCPP_RING_REVERSED) {      // ring lists 0, 3, 2, 1
            const int src = i == 1 ? 3 : (i == 3 ? 1 : i);
            const double tx = qget(cx, src), ty = qget(cy, src);
            cx = tx; cy = ty;
        }
        const int cur = (sci + i) & 3, nxt = (sci + i + 1) & 3;
        const double ccurx = qget(cx, cur), ccury = qget(cy, cur), cnxx = qget(cx, nxt), cnxy = qget(cy, nxt);
        const uint32_t lpw = FCPP_FLAG_HEADLAND | ((uint32_t)(loop * 8) << FCPP_INDEX_SHIFT);
        int64_t ns = 20;
        if (ds > 0) ns = n_for_length(fc_hypot(cnxx - ccurx, cnxy - ccury), ds);
        // failures, in the order plan_field_t meets them: the loop's inset, then side by side the counts, the gap decision
        int fcode = 0;                                       // 1: FCPP_EHEADLAND, 2: FCPP_EUNSUPPORTED
        if (act) {
            if (!inset_ok) fcode = 1;
            else if (ns > INT32_MAX || nt > INT32_MAX) fcode = 1;
        }
        // the turn at this side's end and its reverse fill
        const double ang_nxt = nxt == 0 ? cang0 : (nxt == 1 ? cang1 : (nxt == 2 ? cang2 : cang3));
        const bool has_turn = act && i < 3;
        const bool want_rev = has_turn && loop == 0 && ang_nxt >= 60;       // MLP:1043
        if (act && fcode == 0 && want_rev && pc.gap_decision < 0) fcode = 2;
        const bool add_rev = want_rev && pc.gap_decision > 0;
        double e1x = 0, e1y = 0, e2x = 0, e2y = 0;
        if (has_turn) {
            if (!cloth) {
                corner_arc_point(nxt, cnxx, cnxy, R, pc.arc_c1, pc.arc_s1, e1x, e1y);
                corner_arc_point(nxt, cnxx, cnxy, R, pc.arc_c2, pc.arc_s2, e2x, e2y);
            } else {
                const int qd = corner_quadrant(nxt);
                cac_world_from_unit(pc.cac_u1x, pc.cac_u1y, cnxx, cnxy, qd, pc.Re_half, e1x, e1y);
                cac_world_from_unit(pc.cac_u2x, pc.cac_u2y, cnxx, cnxy, qd, pc.Re_half, e2x, e2y);
            }
        }
        double rdx = -1.0, rdy = 0.0, rlen = 0.0;
        int64_t nr = 0;
        if (add_rev) {
            const double tx = e1x - e2x, ty = e1y - e2y;
            const double nrm = sqrt(tx * tx + ty * ty);
            if (nrm > 1e-6) { rdx = -tx / nrm; rdy = -ty / nrm; }
            rlen = distance_to_boundary(e1x, e1y, rdx, rdy, L, H, R);
            if (ds > 0) nr = n_for_length(rlen, ds);
            else { nr = (int64_t)(rlen / 0.5); if (nr < 10) nr = 10; }
        }
        if (l0 == 0) PSTAMP(7);
        // the first failure in path order decides (rows of a chunk are loops in order, lanes of a quad sides in order); within a side the
        // order is inset (whole loop: its lane 0 first), counts, gap -- one code per lane, the lowest lane wins
        {
            const unsigned bad = rballot(fcode != 0);
            if (bad != 0u) {
                const int first = __builtin_ctz(bad);
                // (a loop whose inset is bad: all four lanes carry code 1, the quad's lane 0 is the first of them)
                const int code = rgeti(fcode, first);
                P16_FAIL(code == 2 ? FCPP_EUNSUPPORTED : FCPP_EHEADLAND);
            }
        }
        // points and primitives of this lane's side, in path order: [ start point (side 0) ] straight [ turn [ reverse fill ] ]
        const int64_t my_pts = act ? ((i == 0 ? 1 : 0) + ns + (has_turn ? nt : 0) + (add_rev ? nr : 0)) : 0;
        const int32_t my_prims = act ? ((i == 0 ? 1 : 0) + 1 + (has_turn ? 1 : 0) + (add_rev ? 1 : 0)) : 0;
        // exclusive prefix over the row's lanes in path order (four steps of a scan inside the row), and the chunk's totals
        int64_t pts_incl = my_pts;
        int32_t prims_incl = my_prims;
        for (int d = 1; d < 16; d <<= 1) {
            const int src = (lane & ~15) | ((l16 - d) & 15);
            const int64_t pv = ((int64_t)__shfl((int)(pts_incl >> 32), src) << 32) | (uint32_t)__shfl((int)(pts_incl & 0xffffffff), src);
            const int32_t qv = __shfl(prims_incl, src);
            if (l16 >= d) { pts_incl += pv; prims_incl += qv; }
        }
        const int64_t pts_before = pts_incl - my_pts;
        const int32_t prims_before = prims_incl - my_prims;
        const int64_t pts_chunk = ((int64_t)rgeti((int)(pts_incl >> 32), 15) << 32) | (uint32_t)rgeti((int)(pts_incl & 0xffffffff), 15);
        const int32_t prims_chunk = rgeti(prims_incl, 15);
        if (l0 == 0) PSTAMP(8);
        const double sxp = qget(cx, sci), syp = qget(cy, sci);          // the loop's start point (a lane move: outside the branches of single lanes)
        if (act) {
            int64_t at = pos + pts_before;
            int32_t pat = prim_pos + prims_before;
            auto put = [&](DevPrim &p) { p.start = at; at += p.n; flag_degenerate(p); if (pat < pc.max_prims) pbase[pat] = p; ++pat; };
            DevPrim p;
            if (i == 0) {
                memset(&p, 0, sizeof(p));
                p.kind = PRIM_POINT; p.n = 1; p.v_nom = veh.max_headland_speed_kmh;
                p.fs = FCPP_KIND_HEAD_START | lpw | ((uint32_t)sci << FCPP_INDEX_SHIFT);
                p.a[0] = sxp; p.a[1] = syp;
                put(p);
            }
            memset(&p, 0, sizeof(p));
            p.kind = PRIM_LINSPACE; p.n = (int32_t)ns; p.v_nom = veh.max_headland_speed_kmh;
            p.fs = FCPP_KIND_HEAD_STRAIGHT | lpw | ((uint32_t)cur << FCPP_INDEX_SHIFT);
            p.a[0] = ccurx; p.a[1] = ccury; p.a[2] = cnxx; p.a[3] = cnxy;
            p.a[4] = lin_step(ccurx, cnxx, ns); p.a[5] = lin_step(ccury, cnxy, ns);
            put(p);
            if (has_turn) {
                memset(&p, 0, sizeof(p));
                p.n = (int32_t)nt; p.v_nom = veh.headland_turn_speed_kmh;
                p.fs = FCPP_KIND_CORNER | lpw | ((uint32_t)nxt << FCPP_INDEX_SHIFT);
                if (!cloth) {
                    p.kind = PRIM_ARC; p.form = nxt;
                    p.a[0] = cnxx; p.a[1] = cnxy; p.a[2] = R; p.a[3] = kHalfPi;
                    p.a[4] = pc.arc_step;
                } else {
                    const int qd = corner_quadrant(nxt);
                    p.kind = PRIM_CAC; p.form = qd;
                    p.a[0] = cnxx; p.a[1] = cnxy; p.a[2] = qd * kHalfPi; p.a[3] = -kHalfPi;
                    p.a[4] = pc.Re_half; p.a[5] = pc.cac_step; p.a[6] = pc.half_T;
                }
                put(p);
                if (add_rev) {
                    memset(&p, 0, sizeof(p));
                    p.kind = PRIM_RAY; p.n = (int32_t)nr; p.v_nom = 2.5;   // MLP:1080
                    p.fs = FCPP_KIND_REVERSE | lpw | ((uint32_t)nxt << FCPP_INDEX_SHIFT);
                    p.a[0] = e1x; p.a[1] = e1y; p.a[2] = rdx; p.a[3] = rdy; p.a[4] = rlen;
                    p.a[5] = lin_step(0.0, rlen, nr);
                    put(p);
                }
            }
        }
        if (l0 == 0) PSTAMP(9);
        // what the row's lane 0 needs of this chunk: the reverse-fill counts (loop 0), the first and the last point of the headland
        if (l0 == 0) {
            const int my_nr = (add_rev && loop == 0) ? (int)nr : 0;
            // the corner c is turned at the end of side j = (c - sci - 1) & 3 of loop 0 (the row's lane j); side 3 has no turn
            const int j0 = (3 - sci) & 3, j1 = (4 - sci) & 3, j2 = (5 - sci) & 3, j3 = (6 - sci) & 3;
            const int v0 = rgeti(my_nr, j0), v1 = rgeti(my_nr, j1), v2 = rgeti(my_nr, j2), v3 = rgeti(my_nr, j3);
            nrev0 = j0 < 3 ? v0 : 0; nrev1 = j1 < 3 ? v1 : 0; nrev2 = j2 < 3 ? v2 : 0; nrev3 = j3 < 3 ? v3 : 0;
            first_head0 = rget(cx, sci); first_head1 = rget(cy, sci);          // loop 0's quad is the row's first
        }
        {
            const int last_l = (num_loops - 1) - l0;         // the last loop's quad in this chunk, if it is here
            if (last_l >= 0 && last_l < 4) { last_head0 = rget(cnxx, last_l * 4 + 3); last_head1 = rget(cnxy, last_l * 4 + 3); }
        }
        pos += pts_chunk;
        prim_pos += prims_chunk;
    }
    PSTAMP(10);
    if (w0) { in.n_reverse[0] = nrev0; in.n_reverse[1] = nrev1; in.n_reverse[2] = nrev2; in.n_reverse[3] = nrev3; }
    if (w0) { in.n_head = pos - n_main; }
    if (w0 && has_start) {   // MLP:437-441
        if (w0) { in.approach_from[0] = f.start_x; in.approach_from[1] = f.start_y; }
        if (w0) { in.approach_to[0] = first_head0; in.approach_to[1] = first_head1; }
    }
    if (w0 && has_end) {     // MLP:443-447
        if (w0) { in.departure_from[0] = last_head0; in.departure_from[1] = last_head1; }
        if (w0) { in.departure_to[0] = f.end_x; in.departure_to[1] = f.end_y; }
    }
    if (w0) { df.n_total = pos; }
    if (w0) { df.prim_count = prim_pos; }
    if (w0) { df.obs_first = (int32_t)f.obstacle_first; df.obs_count = f.n_obstacles; }
    if (lsx < lex) {
        // layer 1's bounding box inside the geofence?  (plan_field_t: four corners x four edges; here corner i = this lane's, edges in order)
        const double slack = cloth ? 1e-3 * R : 0.0;
        const double ext = pc.uturn_dx + slack, hgt = pc.uturn_h + slack;
        const double bx0 = lsx - ext, bx1 = lex + ext, by0 = min_y, by1 = min_y + (double)(P - 1) * W + hgt;
        const double margin = 1e-7 - opt.geofence_tol;
        // (corner order of the host's loops: cxi major, cyi minor -- any order gives the same conjunction)
        double px = (i >> 1) ? bx1 : bx0, py = (i & 1) ? by1 : by0;
        if (rotated) { const double tx = px - ccx, ty = py - ccy; px = (tx * rc - ty * rs) + ccx; py = (tx * rs + ty * rc) + ccy; }
        const double m = margin + 5.684341886080802e-14 * (fabs(px) + fabs(py));
        bool inb = true;
        if (!(gex0 * px + gey0 * py + geo0 >= m)) inb = false;
        if (!(gex1 * px + gey1 * py + geo1 >= m)) inb = false;
        if (!(gex2 * px + gey2 * py + geo2 >= m)) inb = false;
        if (!(gex3 * px + gey3 * py + geo3 >= m)) inb = false;
        const bool all_in = qballot(inb) == 0xFu;
        if (w0) df.span_inside = all_in ? 1 : 0;
    }
    finish(pos, prim_pos, prim_pos);
    PSTAMP(11);
#undef P16_FAIL
}

// a column's total to the host's copy; the flags (generation numbers, see PlanFlag) with the first column of the last scan
__device__ __forceinline__ void publish_total(const int64_t *totals, int64_t *mirror, int col, int64_t value, bool flags, bool over_elsewhere = false)
{
    if (!mirror) return;
    mirror[col] = value;
    // (over_elsewhere: PF_OVER_CAPACITY may still be raised by the workgroup of PC_SPAN in this very launch -- that one publishes it)
    if (flags) for (int k = 0; k < PF_COUNT; ++k) if (!(over_elsewhere && k == PF_OVER_CAPACITY)) mirror[PC_COLS + k] = totals[PC_COLS + k];
}

// the last scan of a counting phase: a column's workgroup has published its total; the last one to arrive tells the host (PX_DONE)
__device__ __forceinline__ void publish_done(int64_t *totals, int64_t *mirror, int n_cols, int64_t done_gen)
{
    if (!mirror || done_gen <= 0) return;
    __threadfence_system();
    const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long *>(totals + PX_ARRIVED), 1ull);
    if (arrived == (unsigned long long)(n_cols - 1)) {
        totals[PX_ARRIVED] = 0;
        __threadfence_system();
        reinterpret_cast<volatile int64_t *>(mirror)[PX_DONE] = done_gen;
    }
}

// ---- exclusive scans of count columns [c0, c1) over the fields -------------------------------------------------------------------------
// inclusive scan of v over the 256 threads of the workgroup; total = the workgroup's sum
__device__ __forceinline__ int64_t wg_incl_scan(int64_t v, int64_t *lds /* 4 */, int64_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    int64_t pre = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int64_t s = lds[q]; if (q < wave) pre += s; tot += s; }
    __syncthreads();
    total = tot;
    return v + pre;
}

// what depends on where a field's span lies in the batch arrays (pt_off): its 512-point chunks, whether its field's own workgroup can write
// it (at most FUSED_SPAN_CHUNKS of them), the alternatives the host chooses between once it has the totals
__device__ __forceinline__ void span_counts(int64_t pt_off, int64_t S, bool is_work, bool fuse_possible, int64_t &c_span, int64_t &c_span_f,
                                            int64_t &c_work_span_pts, int64_t &c_unfusable)
{
    c_span = c_span_f = c_work_span_pts = c_unfusable = 0;
    if (S <= 0) return;
    const int64_t n_chunks = ((pt_off % TILE_POINTS) + S + TILE_POINTS - 1) / TILE_POINTS;
    const bool fusable = is_work && n_chunks <= FUSED_SPAN_CHUNKS;
    const int64_t fused = (fusable && fuse_possible) ? S : 0;
    c_span = n_chunks; c_span_f = fused > 0 ? 0 : n_chunks; c_work_span_pts = fused; c_unfusable = (is_work && !fusable) ? 1 : 0;
}

// phase A: sums of blocks of 1024 fields; grid (blocks, columns)
__global__ __launch_bounds__(256) void k_scan_block_sums(int64_t n, int c0, const int64_t *__restrict__ counts, int64_t *__restrict__ blk_sums, int64_t nblk)
{
    __shared__ int64_t lds[4];
    const int col = c0 + blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * 1024;
    int64_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int64_t i = base + k * 256 + threadIdx.x; if (i < n) v += counts[(int64_t)col * n + i]; }
    int64_t tot;
    wg_incl_scan(v, lds, tot);
    if (threadIdx.x == 0) blk_sums[(int64_t)col * nblk + blockIdx.x] = tot;
}
// phase B: one workgroup per column scans the block sums in place (exclusive) and writes the column's total
__global__ __launch_bounds__(256) void k_scan_block_bases(int c0, int64_t *__restrict__ blk_sums, int64_t nblk, int64_t *__restrict__ totals,
                                                          int64_t *__restrict__ mirror, int with_flags, int64_t done_gen)
{
    __shared__ int64_t lds[4];
    const int col = c0 + blockIdx.x;
    int64_t carry = 0;
    for (int64_t b0 = 0; b0 < nblk; b0 += 256) {
        const int64_t i = b0 + threadIdx.x;
        const int64_t v = i < nblk ? blk_sums[(int64_t)col * nblk + i] : 0;
        int64_t tot;
        const int64_t inc = wg_incl_scan(v, lds, tot);
        if (i < nblk) blk_sums[(int64_t)col * nblk + i] = carry + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) { totals[col] = carry; publish_total(totals, mirror, col, carry, with_flags && blockIdx.x == 0); if (with_flags) publish_done(totals, mirror, (int)gridDim.x, done_gen); }
}
// phase C: exclusive scan inside each block of 1024 fields + the block's base; grid (blocks, columns)
// derive: the column of the points also makes the columns that depend on the fields' point offsets (span_counts), from the offsets it has just
// computed -- the counting pass of a large batch, run in chunks beside the planner, goes without them as a small batch's does
__global__ __launch_bounds__(256) void k_scan_apply(int64_t n, int c0, int64_t *__restrict__ counts, const int64_t *__restrict__ blk_sums,
                                                    int64_t nblk, int64_t *__restrict__ bases, int derive, int fuse_possible)
{
    __shared__ int64_t lds[4];
    const int col = c0 + blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;      // four consecutive fields per thread
    int64_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? counts[(int64_t)col * n + base + k] : 0; s += v[k]; }
    int64_t tot;
    int64_t run = wg_incl_scan(s, lds, tot) - s + blk_sums[(int64_t)col * nblk + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) {
            bases[(int64_t)col * n + base + k] = run;
            if (derive && col == PC_POINTS) {
                const int64_t f = base + k;
                int64_t c_span, c_span_f, c_wsp, c_unf;
                span_counts(run, counts[(int64_t)PC_SPAN_PTS * n + f], counts[(int64_t)PC_WORK * n + f] != 0, fuse_possible != 0, c_span, c_span_f, c_wsp, c_unf);
                counts[(int64_t)PC_SPAN * n + f] = c_span; counts[(int64_t)PC_SPAN_F * n + f] = c_span_f;
                counts[(int64_t)PC_WORK_SPAN_PTS * n + f] = c_wsp; counts[(int64_t)PC_UNFUSABLE * n + f] = c_unf;
            }
        }
        run += v[k];
    }
}

// small batches: one workgroup of 1024 threads per column walks the fields in chunks of 4096 -- one launch instead of three, and one
// pass for the headline's 4096 fields.  derive: the columns that depend on the fields' point offsets (span_counts) are made here, by
// their own workgroups, from the offsets (a scan of PC_POINTS of their own), the spans' lengths (PC_SPAN_PTS) and PC_WORK.
__global__ __launch_bounds__(1024) void k_scan_small(int64_t n, int c0, int64_t *__restrict__ counts, int64_t *__restrict__ bases, int64_t *__restrict__ totals,
                                                     int64_t *__restrict__ mirror, int with_flags, int derive, int fuse_possible, int64_t spec_gen, int64_t done_gen)
{
    __shared__ int64_t lds[16];
    int over = 0;                        // (spec_gen > 0: a speculative setup -- a span of more chunks than its layout has room for raises PF_OVER_CAPACITY)
    const int col = c0 + blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // exclusive scan over the workgroup's 4096 elements (four a thread): -> the base of the thread's first, tot = the chunk's sum
    auto scan4 = [&](const int64_t (&v)[4], int64_t &tot) -> int64_t {
        const int64_t s = v[0] + v[1] + v[2] + v[3];
        int64_t inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) lds[wave] = inc;
        __syncthreads();
        int64_t pre = 0, t = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int64_t x = lds[q]; if (q < wave) pre += x; t += x; }
        __syncthreads();
        tot = t;
        return inc + pre - s;
    };
    const bool derived = derive && (col == PC_SPAN || col == PC_SPAN_F || col == PC_WORK_SPAN_PTS || col == PC_UNFUSABLE);
    int64_t carry = 0, carry_pts = 0;
    for (int64_t b0 = 0; b0 < n; b0 += 4096) {
        const int64_t base = b0 + (int64_t)threadIdx.x * 4;
        int64_t v[4];
        if (derived) {
            int64_t pts[4], S[4], work[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = base + k < n;
                pts[k] = in ? counts[(int64_t)PC_POINTS * n + base + k] : 0;
                S[k] = in ? counts[(int64_t)PC_SPAN_PTS * n + base + k] : 0;
                work[k] = in ? counts[(int64_t)PC_WORK * n + base + k] : 0;
            }
            int64_t tot_pts;
            int64_t off = scan4(pts, tot_pts) + carry_pts;
            carry_pts += tot_pts;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int64_t c_span, c_span_f, c_wsp, c_unf;
                span_counts(off, S[k], work[k] != 0, fuse_possible != 0, c_span, c_span_f, c_wsp, c_unf);
                v[k] = col == PC_SPAN ? c_span : (col == PC_SPAN_F ? c_span_f : (col == PC_WORK_SPAN_PTS ? c_wsp : c_unf));
                if (col == PC_SPAN && c_span > SPEC_SPAN_CHUNKS) over = 1;
                if (base + k < n) counts[(int64_t)col * n + base + k] = v[k];
                off += pts[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = base + k < n ? counts[(int64_t)col * n + base + k] : 0;
        }
        int64_t tot;
        int64_t run = scan4(v, tot) + carry;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (base + k < n) bases[(int64_t)col * n + base + k] = run; run += v[k]; }
        carry += tot;
    }
    const bool spec_span = derive && spec_gen > 0 && col == PC_SPAN;
    if (spec_span) over = __syncthreads_or(over);
    if (threadIdx.x == 0) {
        totals[col] = carry;
        publish_total(totals, mirror, col, carry, with_flags && blockIdx.x == 0, derive && spec_gen > 0);
        if (spec_span) {
            unsigned long long *flag = reinterpret_cast<unsigned long long *>(totals + PC_COLS + PF_OVER_CAPACITY);
            if (over) atomicMax(flag, (unsigned long long)spec_gen);
            const int64_t v = over ? spec_gen : totals[PC_COLS + PF_OVER_CAPACITY];      // (the counting pass may have raised it)
            if (mirror) mirror[PC_COLS + PF_OVER_CAPACITY] = v;
        }
        if (with_flags) publish_done(totals, mirror, (int)gridDim.x, done_gen);
    }
}

// mirror: the totals' copy in the host's pinned memory (or null), written by the scans themselves -- no copy command behind them;
// with_flags: the flags the earlier kernels raised go along (the last scan of the counting phase)
int launch_scan(hipStream_t st, int64_t n, int c0, int c1, const DevPlanScratch &s, int64_t *mirror, int with_flags, int derive = 0, int fuse_possible = 0,
                int64_t spec_gen = 0, int64_t done_gen = 0)
{
    const int64_t nblk = (n + 1023) / 1024;
    const int nc = c1 - c0;
    if (nblk <= devplan_small_blocks()) {
        hipLaunchKernelGGL(k_scan_small, dim3((unsigned)nc), dim3(1024), 0, st, n, c0, s.counts, s.bases, s.totals, mirror, with_flags, derive, fuse_possible, spec_gen, done_gen);
        const hipError_t e0 = hipGetLastError();
        return e0 == hipSuccess ? 0 : (int)e0;
    }
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nblk, (unsigned)nc), dim3(256), 0, st, n, c0, s.counts, s.blk_sums, nblk);
    hipLaunchKernelGGL(k_scan_block_bases, dim3((unsigned)nc), dim3(256), 0, st, c0, s.blk_sums, nblk, s.totals, mirror, with_flags, done_gen);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nblk, (unsigned)nc), dim3(256), 0, st, n, c0, s.counts, s.blk_sums, nblk, s.bases, derive, fuse_possible);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ---- k_tile_fields: one wavefront per field, the tiler's cut at the reference's sampling -------------------------------------------------
// What FieldTiler::tile_field + derive_field (fcpp_tiler.cpp) decide for a field whose lines have 2 points, whose turns 20 and whose
// headland straights 20 (sample_spacing = 0): no straight has a quiet zone, so the field is
//     [ span: all complete passes, closed form ] [ general stretch: the last line + layer 2 ]
// (or one general stretch when the turns are not closed form), the general stretch cut into wave tiles with halos sized from the
// path's own step lengths -- the points evaluated lane-parallel into LDS, the greedy cut walked wave-uniformly -- or, when a wave tile
// cannot be formed, into general tiles of up to 512 points.
// Diagnostic build only (-DFCPP_DIAG_TILE: `make diag-tile`, tools/diag_tile.py; never shipped): 10 ns time stamps of the phases of field
// 1000's counting pass.
#ifdef FCPP_DIAG_TILE
__device__ unsigned long long g_tile_stamps[40], g_fill_stamps[40];
#define TSTAMP(k) do { if (!FILL && field == 1000 && lane == 0 && (k) < 40) g_tile_stamps[k] = wall_clock64(); } while (0)
#define FSTAMP(k) do { if (FILL && field == 1000 && lane == 0 && (k) < 40) g_fill_stamps[k] = wall_clock64(); } while (0)
#else
#define TSTAMP(k) do { } while (0)
#define FSTAMP(k) do { } while (0)
#endif
constexpr int TW_WAVES = 2;                                             // fields per workgroup
constexpr int TW_NW = DEVPLAN_WINDOW;                                   // points of the LDS window that slides along a general stretch
constexpr int TW_LDS_PRIMS = 32;                                        // primitives of a field staged in LDS (more: read where they lie)
constexpr int TW_LDS_TMPL = 36;                                         // turn template samples staged in LDS (U-turn + corner; more: ditto)
template <bool STAGE>
struct TileWaveLds {
    double d[TW_NW];              // d[i - lo] = |p_i - p_(i-1)|   (the window cut; the closed-form cut keeps its records of the field's primitives
                                  // and its copies of the turn templates in the same bytes: cut_rows / cut_tmpl below)
    static_assert(sizeof(double) * TW_NW >= sizeof(CutPrim) * CUT_PRIMS_MAX + sizeof(Pt2) * TW_LDS_TMPL, "the closed-form cut's LDS fits the window's");
    __device__ CutPrim *cut_rows() { return reinterpret_cast<CutPrim *>(d); }
    __device__ Pt2 *cut_tmpl() { return reinterpret_cast<Pt2 *>(reinterpret_cast<unsigned char *>(d) + sizeof(CutPrim) * CUT_PRIMS_MAX); }
    // the field's primitives and the batch's turn templates, staged once: a window point's evaluation then waits on LDS, not on a
    // dependent read from device memory per 64 points (22 of the counting pass's 72 thousand cycles per field)
    unsigned long long prim_words[STAGE ? TW_LDS_PRIMS * (sizeof(DevPrim) / 8) : 1];
    Pt2 tmpl[STAGE ? TW_LDS_TMPL : 1];
    int32_t pstart[DEVPLAN_PRIMS_CAP + 1];   // start of primitive k relative to n_main
    uint8_t pidx[TW_NW];          // primitive (index within the field) of window point w; 0 in layer 1
    uint8_t ins[TW_NW];           // window point w lies inside the geofence with the host's margin
};

__device__ __forceinline__ int32_t clampi(int64_t v) { return (int32_t)(v < -2 ? -2 : (v > ((int64_t)1 << 30) ? ((int64_t)1 << 30) : v)); }

// DENSE: the instance for batches at dense sampling (the dense block and the window cut; no closed-form cut) -- an instance of its own so that
// the sparse instances keep their registers (with the dense block compiled in, the counting pass of large batches went from 79 to 108
// vector registers: four wavefronts per SIMD instead of six)
template <bool FILL, bool STAGE, bool DENSE>
__device__ __forceinline__ void tile_fields_body(int64_t n, const DevTileConsts &tc, const DevConst &cst, const DevField *__restrict__ ftmp, const DevPrim *__restrict__ ptmp,
                                                 fcpp_field_info *__restrict__ info, int64_t *__restrict__ counts,
                                                 const int64_t *__restrict__ bases, int64_t *__restrict__ totals,
                                                 DevTile *__restrict__ keep_tiles, DevWaveTile *__restrict__ keep_wtiles, const DevPlanTables &T)
{
    __shared__ TileWaveLds<STAGE> lds_all[TW_WAVES];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t field = (FILL ? 0 : tc.f0) + (int64_t)blockIdx.x * TW_WAVES + wave;      // (the counting pass of a large batch runs in chunks: fields [f0, f1))
    if (field >= (FILL ? n : tc.f1)) return;
    TSTAMP(0);
    FSTAMP(0);
    TileWaveLds<STAGE> &L = lds_all[wave];
    const DevField &F = ftmp[field];
    const DevPrim *prims = ptmp + field * tc.max_prims;
    const int64_t n_total = F.n_total;
    auto base_of = [&](int col) -> int64_t { return bases[(int64_t)col * n + field]; };
    // a speculative fill pass does nothing when the counting phase raised a flag: the host sets the batch up again (or elsewhere)
    if (FILL && tc.speculative) {
        bool up = false;
        for (int k = 0; k < PF_COUNT; ++k) up = up || totals[PC_COLS + k] == tc.gen;
        if (up) return;
    }
    // (speculative: the fusing of spans is the host's rule applied here -- all fields of field work have fusable spans, or nothing is fused)
    const bool fuse_spans = (FILL && tc.speculative) ? (tc.fuse_spans != 0 && totals[PC_UNFUSABLE] == 0 && totals[PC_WORK_SPAN_PTS] > 0) : tc.fuse_spans != 0;
    // (scanned before either pass -- except before the counting pass of a small batch: tc.no_bases, see span_counts)
    const bool no_bases = !FILL && tc.no_bases != 0;
    const int64_t pt_off = no_bases ? 0 : base_of(PC_POINTS);

    // the field's counts (count pass) / positions (fill pass)
    int64_t c_tiles = 0, c_wave = 0, c_general = 0, c_stat = 0, c_span = 0, c_work = 0, c_open = 0, c_runs = 0, c_span_pts = 0, c_wave_pts = 0,
            c_work_wave_pts = 0, c_wave_inside = 0, c_work_span_pts = 0, c_span_f = 0, c_unfusable = 0, c_chunks_out = 0, c_chunk_pts_out = 0;
    bool dense_entries = false;          // (fill pass: the dense block has written the field's entries, slots and chunks itself -- all but the span's)
    int cls = 0;
    int64_t wave_base = 0, general_base = 0, stat_base = 0, span_base = 0, prim_base = 0;
    if (FILL) {
        wave_base = base_of(PC_WAVE); general_base = base_of(PC_GENERAL); stat_base = base_of(PC_STAT);
        span_base = base_of(fuse_spans ? PC_SPAN_F : PC_SPAN); prim_base = base_of(PC_PRIMS);
    }
    const int prim_count = F.prim_count;
    const int64_t prim_index0 = prim_base;           // batch-wide index of the field's first primitive (fill pass)
    // fill pass: everything it copies out of the scratch -- the field's descriptor, its fcpp_field_info, its primitives, a word per lane and
    // round -- is requested here, before the cut: the copies further down then wait for nothing (they were a chain of six round trips)
    constexpr int NFW = (int)(sizeof(DevField) / 8), NIW = (int)(sizeof(fcpp_field_info) / 8), PWD = (int)(sizeof(DevPrim) / 8), PF_ROUNDS = 7;
    static_assert(sizeof(DevField) % 8 == 0 && sizeof(DevPrim) % 8 == 0 && sizeof(fcpp_field_info) % 8 == 0 && NFW <= 64 && NIW <= 64, "copied as 8-byte words, a lane each");
    unsigned long long fw = 0, iw = 0, pw[PF_ROUNDS];
    const int nwords_p = prim_count * PWD;
    if (FILL) {
        if (lane < NFW) fw = reinterpret_cast<const unsigned long long *>(&F)[lane];
        if (lane < NIW) iw = reinterpret_cast<const unsigned long long *>(&info[field])[lane];
#pragma unroll
        for (int j = 0; j < PF_ROUNDS; ++j) { const int k = lane + 64 * j; pw[j] = k < nwords_p ? reinterpret_cast<const unsigned long long *>(prims)[k] : 0ull; }
        // (pt_off and prim_first become batch-wide: patched by the lane that holds their word)
        constexpr int W_PT = (int)(offsetof(DevField, pt_off) / 8), W_PF = (int)(offsetof(DevField, prim_first) / 8);
        constexpr bool PF_HI = (offsetof(DevField, prim_first) % 8) != 0;
        if (lane == W_PT) fw = (unsigned long long)pt_off;
        if (lane == (int)(offsetof(fcpp_field_info, point_offset) / 8)) iw = (unsigned long long)pt_off;      // (the planner left it 0)
        if (lane == W_PF) fw = PF_HI ? ((fw & 0xffffffffull) | ((unsigned long long)(uint32_t)prim_base << 32)) : ((fw & 0xffffffff00000000ull) | (unsigned long long)(uint32_t)prim_base);
    }

    bool fallback = false, is_work = false;
    TSTAMP(1);
    FSTAMP(1);
    int64_t S = 0, span_k = 0, fused_span = 0;
    int64_t n_wave = 0, n_general = 0;
    // fill pass, lane t: the field's t-th wave tile as written (t < DEVPLAN_KEEP_TILES), its first primitive within the field and the
    // number of primitives its points lie in (0: none of layer 2) -- what the field's pack takes (DevFieldPack)
    DevWaveTile my_wt;
    memset(&my_wt, 0, sizeof my_wt);
    int my_p0_rel = 0, my_np = 0;
    if (n_total > 0) {
        const int64_t per = (int64_t)F.n_line + F.n_turn, P = F.P, gen_main = F.gen_main;
        const double line_step_len = fabs(F.line_step);
        const bool turn_quiet = tc.turn_quiet && F.n_turn == tc.nu && F.line_step > 0.0;
        const bool wave_ok = F.n_turn == tc.nu && (double)tc.wave_factor * tc.two_a * line_step_len >= tc.u_cap;
        // (sample_spacing = 0: 2 points per line, 20 per headland straight -- no quiet zone anywhere; dense sampling: the block below)
        constexpr bool dense = DENSE;
        if ((!dense && F.n_line >= 64) || prim_count > (dense ? 62 : DEVPLAN_PRIMS_CAP)) fallback = true;
        int64_t pos = 0;
        const int64_t need1 = tiler_need_for(tc.c_line, line_step_len, tc.two_a);
        if (need1 >= 0 && per > 0 && gen_main > 0) {
            const bool span = turn_quiet && P >= 2 && (int64_t)F.n_line - need1 < (F.obs_count > 0 ? (int64_t)64 : tc.span_line_max) && (P - 1) * per < (int64_t)0x7fffffff;
            if (span) { S = (P - 1) * per; span_k = (S + (TILE_POINTS - 2) - 1) / (TILE_POINTS - 2); pos = S; }
        }
        // ---- the WINDOW cut of one general stretch [a, b) into wave tiles (FieldTiler::wave_tiles): the sparse field's one stretch (fields the
        // closed-form cut does not take) and, round 5, every stretch of a dense field.  mode 0: counting pass of a sparse field (its first
        // DEVPLAN_KEEP_TILES tiles kept), 1: its fill pass, 2: a dense field's counting pass (nothing written), 3: its fill pass; e_first / w_first:
        // the stretch's first statistics entry / wave tile within the field.  -> tiles cut, or -1: the stretch is the general kernel's.
        const double cap = tiler_halo_cap(tc.u_cap);
        const int64_t n_main = F.n_main;
        bool win_staged = false;
        const DevPrim *wprims = prims;
        const Pt2 *wtu = tc.tu, *wtc = tc.tc;
        int32_t my_pstart = INT32_MAX;
        auto stage_window = [&]() {
            if (win_staged) return;
            win_staged = true;
            const bool lds_prims = STAGE && prim_count <= TW_LDS_PRIMS, lds_tmpl = STAGE && tc.nu + tc.nc <= TW_LDS_TMPL;
            if (lds_prims && lds_tmpl) {
                // the usual field: starts, records and templates requested together, then stored -- one round trip to memory, not three
                static_assert(TW_LDS_PRIMS <= 64 && TW_LDS_TMPL <= 64 && TW_LDS_PRIMS * (sizeof(DevPrim) / 8) <= 6 * 64, "a lane each / six words a lane");
                const unsigned long long *src = reinterpret_cast<const unsigned long long *>(prims);
                const int nw = prim_count * (int)(sizeof(DevPrim) / 8);
                const int64_t st = lane < prim_count ? prims[lane].start : 0;
                unsigned long long w6[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) { const int k = lane + 64 * j; w6[j] = k < nw ? src[k] : 0ull; }
                Pt2 tp = { 0.0, 0.0 };
                if (lane < tc.nu) tp = tc.tu[lane];
                else if (lane < tc.nu + tc.nc) tp = tc.tc[lane - tc.nu];
                if (lane < prim_count) L.pstart[lane] = (int32_t)(st - n_main);
#pragma unroll
                for (int j = 0; j < 6; ++j) { const int k = lane + 64 * j; if (k < nw) L.prim_words[k] = w6[j]; }
                if (lane < tc.nu + tc.nc) L.tmpl[lane] = tp;
            } else {
                for (int k = lane; k < prim_count; k += 64) L.pstart[k] = (int32_t)(prims[k].start - n_main);
                if (lds_prims) {
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(prims);
                    for (int k = lane; k < prim_count * (int)(sizeof(DevPrim) / 8); k += 64) L.prim_words[k] = src[k];
                }
                if (lds_tmpl) {
                    for (int k = lane; k < tc.nu; k += 64) L.tmpl[k] = tc.tu[k];
                    for (int k = lane; k < tc.nc; k += 64) L.tmpl[tc.nu + k] = tc.tc[k];
                }
            }
            wprims = lds_prims ? reinterpret_cast<const DevPrim *>(L.prim_words) : prims;
            wtu = lds_tmpl ? L.tmpl : tc.tu; wtc = lds_tmpl ? L.tmpl + tc.nu : tc.tc;
            wave_sync();                                     // the starts (and the staged records) are in LDS
            my_pstart = lane < prim_count ? L.pstart[lane] : INT32_MAX;       // (fields of at most 64 primitives: their starts, a lane each)
            TSTAMP(2);
        };
        auto cut_stretch = [&](const int64_t a, const int64_t b, const int mode, const int64_t e_first, const int64_t w_first, int64_t &wave_pts,
                               int64_t &inside_cnt) -> int64_t {
            stage_window();
            // the host evaluates the points [lo - 1, hi) of the stretch at once; here a window of TW_NW points slides along with the cut
            // (a tile starting at s touches the points [s - 43, s + 171) at most), the points are the same function values wherever the
            // window lies
            const int64_t lo_all = (a - WAVE_HALO_MAX - 2 > 1) ? a - WAVE_HALO_MAX - 2 : 1;
            const int64_t hi_all = (b + WAVE_HALO_MAX + 2 < n_total) ? b + WAVE_HALO_MAX + 2 : n_total;
            int64_t win0 = 0, win1 = -1;                     // window = path points [win0, win1)
            auto dist = [&](int64_t i) -> double { return L.d[i - win0 - 1]; };
            auto prim_of = [&](int64_t i) -> int { return (int)L.pidx[i - win0]; };
            const int WAVE_LANES = 128;
            int64_t s = a;
            int64_t ordinal = 0;
            bool refused_l = false;
            while (s < b) {
                const int64_t need_lo = (s - WAVE_HALO_MAX - 3 > lo_all - 1) ? s - WAVE_HALO_MAX - 3 : lo_all - 1;
                const int64_t need_hi = (s + WAVE_LANES + WAVE_HALO_MAX + 3 < hi_all) ? s + WAVE_LANES + WAVE_HALO_MAX + 3 : hi_all;
                if (!(win0 <= need_lo && need_hi <= win1)) {
                    // ---- the window's points, lane-parallel: step lengths, geofence margin, primitive of every point
                    wave_sync();
                    win0 = need_lo; win1 = (win0 + TW_NW < hi_all) ? win0 + TW_NW : hi_all;
                    const int nwin = (int)(win1 - win0);
                    double cx = 0.0, cy = 0.0;               // the previous round's last point
                    for (int w_base = 0; w_base < nwin; w_base += 64) {
                        const int w = w_base + lane;
                        const bool valid = w < nwin;
                        const int64_t i = win0 + (valid ? w : nwin - 1);
                        double px = 0.0, py = 0.0;
                        int pk = 0;
#ifdef FCPP_DIAG_TILE
#define WSTAMP(k) do { if (w_base == 128) TSTAMP(k); } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif
                        WSTAMP(24);
                        const int64_t i_last = win0 + ((w_base + 63 < nwin) ? w_base + 63 : nwin - 1);      // (wave-uniform)
                        if (i < gen_main) {
                            int64_t idx, off;
                            if (i_last < ((int64_t)1 << 31)) { const uint32_t q32 = (uint32_t)i / (uint32_t)per; idx = q32; off = (int64_t)((uint32_t)i - q32 * (uint32_t)per); }
                            else { idx = i / per; off = i - idx * per; }
                            tiler_point_main(F, wtu, idx, off, px, py);
                        }
                        if (i_last >= gen_main) {
                            // layer 2: the primitives this round's points lie in, one after the other (wave-uniform: a handful per
                            // round) -- every lane's primitive is the last one that starts at or before its point, as a search per lane finds it
                            const int64_t i_first = win0 + w_base;
                            const int32_t rel = (int32_t)(i - n_main);
                            const int32_t rel_lo = i_first > n_main ? (int32_t)(i_first - n_main) : 0, rel_hi = (int32_t)(i_last - n_main);
                            // every lane's primitive = the last one that starts at or before its point: the first point's by one ballot
                            // over the starts (a lane each), the few that start inside the round counted in; fields of more than 64
                            // primitives search per lane
                            if (prim_count <= 64) {
                                int k = __popcll(__ballot(my_pstart <= rel_lo)) - 1;
                                if (k < 0) k = 0;
                                pk = k;
                                for (++k; k < prim_count; ++k) {
                                    const int32_t st_k = __builtin_amdgcn_readlane(my_pstart, k);
                                    if (st_k > rel_hi) break;
                                    pk += rel >= st_k ? 1 : 0;
                                }
                            } else {
                                int lo_k = 0, hi_k = prim_count - 1;
                                while (lo_k < hi_k) { const int m = (lo_k + hi_k + 1) >> 1; if (L.pstart[m] <= rel) lo_k = m; else hi_k = m - 1; }
                                pk = lo_k;
                            }
                            WSTAMP(25);
                            if (i >= gen_main) {
                                const DevPrim q = wprims[pk];
                                tiler_point_prim(q, wtu, wtc, (int)(i - q.start), px, py);
                            } else pk = 0;
                        }
                        WSTAMP(26);
                        double qx = __shfl_up(px, 1), qy = __shfl_up(py, 1);
                        if (lane == 0) { qx = cx; qy = cy; }
                        cx = __shfl(px, 63); cy = __shfl(py, 63);
                        if (valid) {
                            if (w >= 1) { const double dx = px - qx, dy = py - qy; L.d[w - 1] = sqrt(dx * dx + dy * dy); }
                            WSTAMP(27);
                            L.pidx[w] = (uint8_t)pk;
                            L.ins[w] = tiler_inside(F, px, py, tc.fence_margin) ? 1 : 0;
                        }
                        WSTAMP(28);
                    }
                    wave_sync();
                }
                // ---- one step of the greedy cut, wave-uniform
                TSTAMP(3 + 4 * (int)ordinal);
                const int Hb = tiler_back_halo(dist, s, tc.two_a, cap);
                TSTAMP(4 + 4 * (int)ordinal);
                if (Hb < 0) { refused_l = true; break; }
                const int64_t cmax = (b - s < WAVE_LANES - Hb) ? b - s : WAVE_LANES - Hb;
                const int64_t first0 = s - Hb;
                const int pa0 = first0 + Hb + cmax - 1 >= gen_main ? prim_of(first0 > gen_main ? first0 : gen_main) : 0;
                // the largest count whose forward halo still fits and whose points lie in at most nine primitives: candidates lane-parallel
                int64_t c = 0;
                int Hf = -1;
                for (int64_t cb = 0; cb < cmax; cb += 64) {
                    const int64_t cand = cmax - cb - lane;
                    bool ok = false;
                    int hf = -1;
                    if (cand >= 1) {
                        hf = tiler_fwd_halo(dist, s + cand - 1, n_total, tc.two_a, cap);
                        ok = hf >= 0 && Hb + cand + hf <= WAVE_LANES;
                        if (ok) { const int64_t last0 = s + cand - 1 + hf; if (last0 >= gen_main && prim_of(last0) - pa0 > 8) ok = false; }
                    }
                    const unsigned long long m = __ballot(ok);
                    if (m) {
                        const int l0 = __builtin_ctzll(m);
                        c = cmax - cb - l0;
                        Hf = __shfl(hf, l0);
                        break;
                    }
                }
                if (c < (b - s < 8 ? b - s : 8)) { refused_l = true; break; }
                TSTAMP(5 + 4 * (int)ordinal);
                const int64_t first = s - Hb, last = s + c - 1 + Hf;
                // every output point inside the geofence with the margin?
                bool all_in = true;
                for (int64_t o0 = 0; o0 < c; o0 += 64) {
                    const int64_t o = o0 + lane;
                    const bool bad = o < c && L.ins[s + o - win0] == 0;
                    if (__ballot(bad)) all_in = false;
                }
                int pa = 0, pb = 0;
                if (last >= gen_main) {
                    const int64_t fl2 = first > gen_main ? first : gen_main;
                    pa = prim_of(fl2); pb = prim_of(last);
                    if (pb - pa > 8) { refused_l = true; break; }
                    bool bad = false;
                    for (int k = pa + 1; k <= pb; ++k) if (L.pstart[k] <= L.pstart[k - 1]) bad = true;
                    if (bad) { refused_l = true; break; }
                }
                // (kept by the counting pass, indices relative to the field: a sparse field's first DEVPLAN_KEEP_TILES tiles, a dense field's first
                // DEVPLAN_KEEP_ROWS over all its stretches -- row = the tile's number within the field)
                const bool wr_fill = mode == 1 || mode == 3;
                const bool wr_keep = (mode == 0 && ordinal < DEVPLAN_KEEP_TILES) || (mode == 2 && w_first + ordinal < DEVPLAN_KEEP_ROWS);
                if ((wr_fill || wr_keep) && lane == 0) {
                    // (count pass: indices relative to the field -- primitive 0 = the field's first, entry 0 = its first, out_base = first)
                    const int64_t p_base = wr_fill ? prim_index0 : 0;
                    DevTile t;
                    t.field = (int32_t)field; t.count = (int32_t)c; t.start = s; t.quiet = 5; t.stat_tile = Hb | (Hf << 16);
                    if (first < gen_main) { t.idx0 = (int32_t)(first / per); t.off0 = (int32_t)(first % per); }
                    else { t.idx0 = (int32_t)(p_base + prim_of(first)); t.off0 = 0; }
                    DevWaveTile wt;
                    memset(&wt, 0, sizeof wt);
                    wt.out_base = (wr_fill ? pt_off : 0) + first; wt.field = (int32_t)field;
                    wt.tile = (int32_t)((wr_fill ? stat_base : 0) + e_first + ordinal);
                    wt.count = (uint8_t)c; wt.hb = (uint8_t)Hb; wt.hf = (uint8_t)Hf; wt.inside = all_in ? 1 : 0;
                    wt.rel_main = clampi(gen_main - first); wt.rel_seam = clampi(n_main - first); wt.rel_last = clampi(n_total - 1 - first);
                    wt.rel_zero = clampi(-first);
                    wt.idx0 = t.idx0; wt.off0 = t.off0;
                    for (int k = 0; k < 8; ++k) wt.thr[k] = 255;
                    if (last >= gen_main) {
                        wt.p0 = (int32_t)(p_base + pa);
                        wt.r0 = (int32_t)(first - (n_main + L.pstart[pa]));
                        for (int k = pa + 1; k <= pb; ++k) wt.thr[k - pa - 1] = (uint8_t)(n_main + L.pstart[k] - first);
                    }
                    if (wr_fill) { T.tiles[stat_base + e_first + ordinal] = t; T.wtiles[wave_base + w_first + ordinal] = wt; }
                    else { keep_tiles[field * DEVPLAN_KEEP_ROWS + w_first + ordinal] = t; keep_wtiles[field * DEVPLAN_KEEP_WROWS + w_first + ordinal] = wt; }
                }
                TSTAMP(6 + 4 * (int)ordinal);
                inside_cnt += all_in ? 1 : 0;
                wave_pts += c;
                ++ordinal;
                s += c;
            }
            return refused_l ? -1 : ordinal;
        };
        // ---- DENSE sampling (round 5; FieldTiler::tile_field + derive_field for a field whose complete passes form a span): behind the span the
        // last swath line and every headland straight have a quiet zone of their own (a run: one tile, one statistics entry, chunks on
        // 512-point boundaries of the batch arrays); the stretches between the zones are cut into wave tiles (samplings coarse enough for
        // them -- 0.18 m and up at the default accelerations -- by the window cut above, a stretch after the other; a stretch it refuses
        // is the general kernel's, as on the host) or into general tiles.  The zones a lane each (lane 0: the last line, lane 1 + k:
        // primitive k), the stretches a lane each, places by prefix sums over the lanes.  Not taken here (PF_FALLBACK: the host sets the
        // batch up): fields without a span (obstacles, turns that are not closed form), more than 62 primitives, fields of field work.
        int64_t n_quiet = 0, c_chunks = 0, c_chunk_pts = 0;
        if (dense && !fallback) {
            if (!(S > 0 && F.obs_count == 0 && gen_main == F.n_main)) fallback = true;
            else {
                int64_t q_s = 0, q_n = 0;
                int q_kind = 0, q_i0 = 0, q_o0 = 0;
                bool valid = false;
                if (lane == 0) {
                    const int64_t Z = (int64_t)F.n_line - need1;          // (no margin at its start behind a span; need1 points before the seam to layer 2)
                    if (Z >= 64) { valid = true; q_s = S; q_n = Z; q_kind = 1; q_i0 = (int)(P - 1); q_o0 = 0; }
                } else if (lane - 1 < prim_count) {
                    const DevPrim &pr = prims[lane - 1];
                    if (pr.kind == PRIM_LINSPACE) {
                        const double ms = pr.v_nom / 3.6;
                        const int64_t need2 = tiler_need_for(ms * ms, sqrt(pr.a[4] * pr.a[4] + pr.a[5] * pr.a[5]), tc.two_a);
                        const int64_t Z = (int64_t)pr.n - 2 * need2;
                        if (need2 >= 0 && Z >= 64) { valid = true; q_s = pr.start + need2; q_n = Z; q_kind = 2; q_i0 = lane - 1; q_o0 = (int)need2; }
                    }
                }
                const unsigned long long vm = __ballot(valid);
                const int nq = __popcll(vm), rk = __popcll(vm & ((1ull << lane) - 1ull));
                // the runs in path order: through the window's bytes into the lanes (lane j: run j and the stretch in front of it)
                int64_t s_j = 0, z_j = 0, a_j = 0, len_j = 0;
                int k_j = 0, i_j = 0, o_j = 0;
                {
                    int64_t *qs = reinterpret_cast<int64_t *>(L.d), *qn = qs + 64;
                    int32_t *qk = reinterpret_cast<int32_t *>(qn + 64), *qi = qk + 64, *qo = qi + 64;
                    static_assert(sizeof(L.d) >= 64 * (8 + 8 + 4 + 4 + 4), "the runs of a dense field fit the window's bytes");
                    wave_sync();
                    if (valid) { qs[rk] = q_s; qn[rk] = q_n; qk[rk] = q_kind; qi[rk] = q_i0; qo[rk] = q_o0; }
                    wave_sync();
                    if (lane < nq) { s_j = qs[lane]; z_j = qn[lane]; k_j = qk[lane]; i_j = qi[lane]; o_j = qo[lane]; }
                    // stretch j = [end of run j - 1 (the span's for j = 0), start of run j (the path's end for j = nq))
                    if (lane <= nq) {
                        a_j = lane == 0 ? S : qs[lane - 1] + qn[lane - 1];
                        const int64_t b_j = lane == nq ? n_total : s_j;
                        len_j = b_j > a_j ? b_j - a_j : 0;
                    }
                    wave_sync();                                      // (the window cut writes these bytes)
                }
                int64_t ng_j = (len_j + TILE_POINTS - 1) / TILE_POINTS, nw_j = 0;       // general tiles of the stretch -- or its wave tiles
                int64_t wave_pts = 0, inside_cnt = 0;
                if (wave_ok) {
                    // a stretch after the other, in path order.  The counting pass keeps the field's first DEVPLAN_KEEP_ROWS wave tiles (indices and
                    // entries relative to the field) and, in the row behind them, a byte per stretch: its wave tiles, or 255 -- the general
                    // kernel's, as the host decides.  The fill pass copies the kept tiles (a lane each); a field with more cuts again, only the
                    // stretches that take wave tiles, whose records it can therefore write as it goes
                    unsigned char *verdicts = reinterpret_cast<unsigned char *>(keep_wtiles + field * DEVPLAN_KEEP_WROWS + DEVPLAN_KEEP_ROWS);
                    unsigned long long refused_mask = 0ull;
                    bool recut = true;
                    if (FILL) {
                        // what the counting pass found: wave tiles per stretch (a lane each).  A field whose wave tiles were all kept is not cut again
                        const int vb = lane <= nq ? (int)verdicts[lane] : 0;
                        refused_mask = __ballot(vb == 255);
                        if (lane <= nq && vb != 255 && len_j > 0) { nw_j = vb; ng_j = 0; }
                        int64_t tot = nw_j;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
                        recut = tot > DEVPLAN_KEEP_ROWS;
                        if (!recut) {
                            if (lane < tot) {
                                DevTile t = keep_tiles[field * DEVPLAN_KEEP_ROWS + lane];
                                DevWaveTile wt = keep_wtiles[field * DEVPLAN_KEEP_WROWS + lane];
                                const int64_t first = wt.out_base;                       // (kept relative to the field)
                                const int nl = (int)wt.hb + wt.count + wt.hf;
                                if (first >= gen_main) { t.idx0 += (int32_t)prim_index0; wt.idx0 = t.idx0; }
                                if (wt.rel_main < nl) wt.p0 += (int32_t)prim_index0;     // the tile holds points of layer 2
                                wt.out_base = pt_off + first;
                                const int64_t e = stat_base + wt.tile;                   // (kept: the entry within the field)
                                wt.tile = (int32_t)e;
                                T.tiles[e] = t;
                                T.wtiles[wave_base + lane] = wt;
                                T.stat_ids[e] = (int32_t)e; T.stat_run[e] = 0;
                                unsigned long long *slot = reinterpret_cast<unsigned long long *>(T.partial + e);
                                for (int w = 0; w < 13; ++w) slot[w] = 0ull;
                            }
                        }
                    }
                    int64_t e_run = 1, w_run = 0;
                    if (recut) {
                        if (FILL && lane <= nq) { nw_j = 0; ng_j = (len_j + TILE_POINTS - 1) / TILE_POINTS; }
                        for (int j = 0; j <= nq; ++j) {
                            const int64_t a_s = __shfl(a_j, j), l_s = __shfl(len_j, j);
                            int64_t t_s = (l_s + TILE_POINTS - 1) / TILE_POINTS;
                            if (l_s > 0 && !((refused_mask >> j) & 1ull)) {
                                const int64_t nt = cut_stretch(a_s, a_s + l_s, FILL ? 3 : 2, e_run, w_run, wave_pts, inside_cnt);
                                if (nt >= 0) {
                                    if (lane == j) { nw_j = nt; ng_j = 0; }
                                    if (FILL)
                                        for (int64_t i = lane; i < nt; i += 64) {
                                            const int64_t e = stat_base + e_run + i;
                                            T.stat_ids[e] = (int32_t)e; T.stat_run[e] = 0;
                                            unsigned long long *slot = reinterpret_cast<unsigned long long *>(T.partial + e);
                                            for (int w = 0; w < 13; ++w) slot[w] = 0ull;
                                        }
                                    t_s = nt; w_run += nt;
                                } else if (FILL) fallback = true;              // (cannot happen: the counting pass cut this stretch with the same code)
                                else refused_mask |= 1ull << j;
                            }
                            e_run += t_s + 1;
                        }
                    }
                    if (!FILL && lane <= nq) {
                        // (a stretch of 255 wave tiles or more: not representable in its byte -- 20 000 points without a quiet zone: the host's)
                        if (nw_j >= 255) fallback = true;
                        verdicts[lane] = ((refused_mask >> lane) & 1ull) ? (unsigned char)255 : (unsigned char)nw_j;
                    }
                    fallback = __ballot(fallback) != 0ull;
                }
                const int64_t t_j = lane <= nq ? nw_j + ng_j : 0;                  // entries of the stretch
                int64_t J_j = 0;
                if (lane < nq) J_j = (((pt_off + s_j) % TILE_POINTS) + z_j + TILE_POINTS - 1) / TILE_POINTS;
                int64_t g_incl = lane <= nq ? ng_j : 0, w_incl = nw_j, j_incl = J_j, z_sum = z_j;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int64_t u = __shfl_up(g_incl, o), v = __shfl_up(w_incl, o), w = __shfl_up(j_incl, o), zz = __shfl_up(z_sum, o);
                    if (lane >= o) { g_incl += u; w_incl += v; j_incl += w; z_sum += zz; }
                }
                if (lane > nq) ng_j = 0;
                const int64_t g_before = g_incl - ng_j, w_before = w_incl - nw_j, j_before = j_incl - J_j;
                n_general = __shfl(g_incl, 63); n_wave = __shfl(w_incl, 63); c_chunks = __shfl(j_incl, 63); c_chunk_pts = __shfl(z_sum, 63);
                c_wave_pts = wave_pts; c_wave_inside = inside_cnt;
                n_quiet = nq;
                const int64_t E_j = 1 + g_before + w_before + lane;                  // the stretch's first entry within the field
                if (FILL && !fallback) {
                    const int64_t chunk_base = base_of(PC_CHUNKS);
                    // general tiles of stretch `lane`: near-equal, at most 512 points (emit_general)
                    if (lane <= nq && ng_j > 0) {
                        const int64_t bs = len_j / ng_j, rem = len_j % ng_j;
                        for (int64_t i = 0; i < ng_j; ++i) {
                            const int64_t st = a_j + i * bs + (i < rem ? i : rem), cnt = bs + (i < rem ? 1 : 0);
                            const int64_t e = stat_base + E_j + i;
                            const bool l1 = per > 0 && st < gen_main;
                            DevTile t;
                            t.field = (int32_t)field; t.start = st; t.count = (int32_t)cnt; t.quiet = 0; t.stat_tile = (int32_t)e;
                            t.idx0 = l1 ? (int32_t)(st / per) : 0; t.off0 = l1 ? (int32_t)(st % per) : 0;
                            T.tiles[e] = t;
                            T.general_ids[general_base + g_before + i] = (int32_t)e;
                            T.stat_ids[e] = (int32_t)e; T.stat_run[e] = 0;
                            unsigned long long *slot = reinterpret_cast<unsigned long long *>(T.partial + e);
                            for (int w = 0; w < 13; ++w) slot[w] = 0ull;
                        }
                    }
                    // the runs: tile, entry, closed-form statistics, chunks
                    if (lane < nq) {
                        const int64_t e = stat_base + E_j + t_j;
                        const int64_t capq = TILE_POINTS - 2, k = (z_j + capq - 1) / capq, bs = z_j / k, rem = z_j % k;
                        DevTile t;
                        t.field = (int32_t)field; t.start = s_j; t.count = (int32_t)(bs + (rem > 0 ? 1 : 0)); t.quiet = k_j; t.stat_tile = 0;
                        t.idx0 = k_j == 2 ? (int32_t)(prim_index0 + i_j) : i_j; t.off0 = o_j;
                        T.tiles[e] = t;
                        T.stat_ids[e] = (int32_t)e; T.stat_run[e] = z_j;
                        double2 junc = make_double2(0.0, 0.0);
                        if (F.P >= 2 && F.n_line >= 2 && F.n_turn >= 1) junc.x = line_start_curvature(F, cst, 1, junc.y);
                        FieldStatView fv;
                        fv.n_line = F.n_line; fv.n_turn = F.n_turn; fv.reverse_order = F.reverse_order; fv.line_step = F.line_step; fv.n_main = F.n_main; fv.junc = junc;
                        DevTile tl = t;
                        tl.idx0 = i_j;                                   // (the field's own primitives: `prims`)
                        const DevRun run = { (int32_t)e, 0, z_j };
                        TilePartial tp = quiet_run_partial(run, tl, fv, prims, cst);
                        tp.n_viol = tp.n_outside = tp.n_in_obstacle = tp.n_adjusted = 0;
                        T.partial[e] = tp;
                        const int64_t g0r = pt_off + s_j;
                        const int64_t c_first = (z_j < TILE_POINTS - (g0r % TILE_POINTS)) ? z_j : TILE_POINTS - (g0r % TILE_POINTS);
                        for (int64_t j = 0; j < J_j; ++j) {
                            const int64_t done = j == 0 ? 0 : c_first + (j - 1) * TILE_POINTS;
                            DevTile ch = t;
                            ch.start = s_j + done; ch.count = (int32_t)(j == 0 ? c_first : ((z_j - done < TILE_POINTS) ? z_j - done : TILE_POINTS));
                            ch.stat_tile = (int32_t)e; ch.off0 = (int32_t)(t.off0 + done);
                            T.chunks[chunk_base + j_before + j] = ch;
                        }
                    }
                }
            }
        }
        const int64_t a = pos, b = n_total, G = b - a;
        bool use_wave = wave_ok && G > 0 && !fallback && !dense;
        // the count pass has already decided whether the stretch takes wave tiles: the fill pass reads its verdict
        if (FILL && use_wave) {
            const int64_t cw = counts[(int64_t)PC_WAVE * n + field];
            if (cw == 0) use_wave = false;
            else if (cw <= DEVPLAN_KEEP_TILES || (tc.closed_cut != 0 && a == cut_span_points(F, tc.cut) && cut_applies(F, tc.cut, a))) {       // (a closed-form cut: every tile was kept)
                // the counting pass kept this field's wave tiles: copy them, indices made batch-wide
                use_wave = false;
                n_wave = cw;
                if (lane < cw) {
                    DevTile t = keep_tiles[field * DEVPLAN_KEEP_ROWS + lane];
                    DevWaveTile wt = keep_wtiles[field * DEVPLAN_KEEP_WROWS + lane];
                    const int64_t first = wt.out_base;                       // (kept relative to the field)
                    const int nl = (int)wt.hb + wt.count + wt.hf;
                    if (first >= gen_main) { t.idx0 += (int32_t)prim_index0; wt.idx0 = t.idx0; }
                    if (wt.rel_main < nl) {                                  // the tile holds points of layer 2
                        my_p0_rel = wt.p0;
                        my_np = 1;
                        for (int k = 0; k < 8; ++k) my_np += wt.thr[k] != 255 ? 1 : 0;
                        wt.p0 += (int32_t)prim_index0;
                    }
                    wt.out_base = pt_off + first;
                    wt.tile += (int32_t)stat_base;
                    T.tiles[stat_base + (span_k > 0 ? 1 : 0) + lane] = t;
                    T.wtiles[wave_base + lane] = wt;
                    my_wt = wt;
                }
            }
        }
        // ---- round 5: the general stretch of a field with a closed-form span is cut IN CLOSED FORM (fcpp_cutfn.h, the function the host
        // tiler runs): the step lengths the halos are sized from are the primitives' own steps and the distances between their end points
        // -- the primitives a lane each, then the tiles of a candidate cut a lane each -- no point of the stretch is evaluated
        if (!DENSE && !FILL && use_wave && tc.closed_cut != 0 && a == cut_span_points(F, tc.cut) && cut_applies(F, tc.cut, a)) {
            use_wave = false;
            const int64_t n_main = F.n_main;
            TSTAMP(10);
            // the templates and their chord tables through LDS (a halo walk reads a chord per step: from device memory every step was a round
            // trip of its own), requested together with the primitives' records
            CutConsts lc = tc.cut;
            const int nt_all = tc.cut.nu + tc.cut.nc;
            const bool cut_staged = nt_all <= TW_LDS_TMPL;
            Pt2 tp = { 0.0, 0.0 };
            if (cut_staged && lane < nt_all) tp = lane < tc.cut.nu ? tc.cut.tu[lane] : tc.cut.tc[lane - tc.cut.nu];
            // 1. the primitives' records
            bool p_ok = true;
            double ex = 0.0, ey = 0.0, fx = 0.0, fy = 0.0;
            DevPrim q;
            memset(&q, 0, sizeof q);
            if (lane < prim_count) q = prims[lane];
            TSTAMP(15);
            if (cut_staged) {
                if (lane < nt_all) L.cut_tmpl()[lane] = tp;
                lc.tu = L.cut_tmpl(); lc.tc = L.cut_tmpl() + tc.cut.nu;
                wave_sync();
            }
            TSTAMP(16);
            if (lane < prim_count) {
                cut_prim_end(q, lc, false, fx, fy);
                TSTAMP(17);
                ex = fx; ey = fy;
                if (q.n > 1) cut_prim_end(q, lc, true, ex, ey);
            }
            TSTAMP(11);
            double px = __shfl_up(ex, 1), py = __shfl_up(ey, 1);          // the point before a primitive's first: its predecessor's last ...
            if (lane == 0) cut_main_end(F, lc, px, py);                    // ... or the last point of layer 1
            if (lane < prim_count) {
                L.cut_rows()[lane] = cut_prim_rec(q, F, lc, px, py, fx, fy, ex, ey, p_ok);
                L.pstart[lane] = (int32_t)(q.start - n_main);
            }
            const bool prims_ok = __ballot(lane < prim_count && !p_ok) == 0ull;
            wave_sync();
            TSTAMP(12);
            // 2. the cut: candidate cuts of T near-equal tiles, the tiles of one a lane each; the decisions of cut_field, taken on the first
            // tile (in path order) that does not fit
            struct CutView { const CutPrim *p; __device__ const CutPrim &operator()(int k) const { return p[k]; } };
            const CutView pv{ L.cut_rows() };
            const double cap = tiler_halo_cap(tc.u_cap);
            const int32_t G32 = (int32_t)G;
            const int32_t cut_start = lane < prim_count ? (int32_t)(q.start - n_main) : INT32_MAX;      // this lane's primitive's first point
            int nt = 0, Hb = 0, Hf = 0;
            int64_t s = 0, c = 0;
            bool in = false;
            if (prims_ok)
                for (int32_t T = cut_first_T(G32); cut_T_possible(G32, T); ++T) {
                    int code = 0;
                    if (lane < T) { s = a + cut_tile_start(G32, T, lane); c = cut_tile_count(G32, T, lane); }
                    // the segments of the points s - 1 and s + c - 1 of every tile: the primitives' starts a lane each, a ballot per tile and point
                    int kj = 0, ke = 0;
                    int32_t rj = 0, re = 0;
                    for (int t = 0; t < T; ++t) {
                        const int64_t j_t = __shfl(s, t) - 1, e_t = __shfl(s + c - 1, t);
                        const int32_t relj = (int32_t)(j_t - n_main), rele = (int32_t)(e_t - n_main);
                        const int a_t = __popcll(__ballot(cut_start <= relj)) - 1, b_t = __popcll(__ballot(cut_start <= rele)) - 1;
                        if (lane == t) { kj = a_t; ke = b_t; }
                    }
                    if (lane < T) {
                        const int64_t Sl = ((int64_t)F.P - 1) * per, j = s - 1, e = s + c - 1;
                        if (j < Sl) { kj = -2; rj = (int32_t)(j - (Sl - F.n_turn)); } else if (j < n_main) { kj = -1; rj = (int32_t)(j - Sl); } else rj = (int32_t)(j - n_main) - L.cut_rows()[kj < 0 ? 0 : kj].start_rel;
                        if (e < Sl) { ke = -2; re = (int32_t)(e - (Sl - F.n_turn)); } else if (e < n_main) { ke = -1; re = (int32_t)(e - Sl); } else re = (int32_t)(e - n_main) - L.cut_rows()[ke < 0 ? 0 : ke].start_rel;
                        code = cut_tile_eval_at(F, lc, pv, prim_count, cap, s, c, kj, rj, ke, re, Hb, Hf, in);
                    }
                    const unsigned long long bad = __ballot(code != 0);
                    if (bad == 0ull) { nt = (int)T; break; }
                    if (__shfl(code, __builtin_ctzll(bad)) == 1) break;           // a halo too long: the general kernel's
                }
            TSTAMP(13);
            if (nt != 0) {
                // the last primitive that starts at or before point i >= n_main: the starts a lane each, one ballot per tile and question
                const int32_t my_start = lane < prim_count ? (int32_t)(q.start - n_main) : INT32_MAX;
                const int64_t first_l = s - Hb, last_l = s + c - 1 + Hf;
                int pa_l = 0, pb_l = 0, pf_l = 0;
                for (int t = 0; t < nt; ++t) {
                    const int64_t f_t = __shfl(first_l, t), l_t = __shfl(last_l, t);
                    const int32_t rf = (int32_t)((f_t > gen_main ? f_t : gen_main) - n_main), rl = (int32_t)(l_t - n_main), r1 = (int32_t)(f_t - n_main);
                    const int a_t = __popcll(__ballot(my_start <= rf)) - 1, b_t = __popcll(__ballot(my_start <= rl)) - 1, f1_t = __popcll(__ballot(my_start <= r1)) - 1;
                    if (lane == t) { pa_l = a_t < 0 ? 0 : a_t; pb_l = b_t < 0 ? 0 : b_t; pf_l = f1_t < 0 ? 0 : f1_t; }
                }
                const int64_t wave_pts = G;
                const int64_t inside_cnt = __popcll(__ballot(lane < nt && in));
                if (lane < nt) {
                    const int64_t first = s - Hb, last = s + c - 1 + Hf;
                    const int64_t p_base = FILL ? prim_index0 : 0;
                    DevTile t;
                    t.field = (int32_t)field; t.count = (int32_t)c; t.start = s; t.quiet = 5; t.stat_tile = Hb | (Hf << 16);
                    if (first < gen_main) { const uint32_t q32 = (uint32_t)first / (uint32_t)per; t.idx0 = (int32_t)q32; t.off0 = (int32_t)((uint32_t)first - q32 * (uint32_t)per); }   // (first < 2^31: cut_applies)
                    else { t.idx0 = (int32_t)(p_base + pf_l); t.off0 = 0; }
                    DevWaveTile wt;
                    memset(&wt, 0, sizeof wt);
                    wt.out_base = (FILL ? pt_off : 0) + first; wt.field = (int32_t)field;
                    wt.tile = (int32_t)((FILL ? stat_base : 0) + (span_k > 0 ? 1 : 0) + lane);
                    wt.count = (uint8_t)c; wt.hb = (uint8_t)Hb; wt.hf = (uint8_t)Hf; wt.inside = in ? 1 : 0;
                    wt.rel_main = clampi(gen_main - first); wt.rel_seam = clampi(n_main - first); wt.rel_last = clampi(n_total - 1 - first);
                    wt.rel_zero = clampi(-first);
                    wt.idx0 = t.idx0; wt.off0 = t.off0;
                    for (int k = 0; k < 8; ++k) wt.thr[k] = 255;
                    if (last >= gen_main) {
                        const int pa = pa_l, pb = pb_l;
                        wt.p0 = (int32_t)(p_base + pa);
                        wt.r0 = (int32_t)(first - (n_main + L.pstart[pa]));
                        for (int k = pa + 1; k <= pb; ++k) wt.thr[k - pa - 1] = (uint8_t)(n_main + L.pstart[k] - first);
                    }
                    if (FILL) { T.tiles[stat_base + (span_k > 0 ? 1 : 0) + lane] = t; T.wtiles[wave_base + lane] = wt; }
                    else { keep_tiles[field * DEVPLAN_KEEP_ROWS + lane] = t; keep_wtiles[field * DEVPLAN_KEEP_WROWS + lane] = wt; }
                }
                n_wave = nt; c_wave_pts = wave_pts; c_wave_inside = inside_cnt;
                TSTAMP(14);
            }
        }
        if (use_wave) {
            int64_t wave_pts = 0, inside_cnt = 0;
            const int64_t nt = cut_stretch(a, b, FILL ? 1 : 0, span_k > 0 ? 1 : 0, 0, wave_pts, inside_cnt);
            if (nt >= 0) { n_wave = nt; c_wave_pts = wave_pts; c_wave_inside = inside_cnt; }
            TSTAMP(36);
        }
        if (G > 0 && n_wave == 0 && !fallback && !dense) {
            // general tiles of at most 512 points, near-equal
            n_general = (G + TILE_POINTS - 1) / TILE_POINTS;
            if (FILL) {
                const int64_t bs = G / n_general, rem = G % n_general;
                const bool in1 = per > 0;
                for (int64_t j = lane; j < n_general; j += 64) {
                    const int64_t st = a + j * bs + (j < rem ? j : rem), cnt = bs + (j < rem ? 1 : 0);
                    const bool l1 = in1 && st < gen_main;
                    DevTile t;
                    t.field = (int32_t)field; t.start = st; t.count = (int32_t)cnt; t.quiet = 0;
                    t.stat_tile = (int32_t)(stat_base + (span_k > 0 ? 1 : 0) + j);
                    t.idx0 = l1 ? (int32_t)(st / per) : 0; t.off0 = l1 ? (int32_t)(st % per) : 0;
                    T.tiles[stat_base + (span_k > 0 ? 1 : 0) + j] = t;
                    T.general_ids[general_base + j] = (int32_t)(stat_base + (span_k > 0 ? 1 : 0) + j);
                }
            }
        }
        // a field whose general points are all in a few wave tiles is planned AND reduced by one workgroup (k_plan_sparse_fields), which
        // then writes the field's span too (fuse_spans): its chunks are not in k_plan_quiet's list
        {
            const int64_t ne0 = (span_k > 0 ? 1 : 0) + n_wave + n_general + n_quiet;
            const int fw_max = tc.field_work_tiles < FIELD_WORK_TILES ? tc.field_work_tiles : FIELD_WORK_TILES;
            is_work = n_general == 0 && n_wave >= 1 && n_wave <= fw_max && ne0 <= FIELD_WORK_ENTRIES;
            // (a dense field of field work -- few wave tiles, nothing general -- gets a pack and a fused span on the host: not built here)
            if (dense && is_work) { is_work = false; fallback = true; }
        }
        // ---- the span's tiles (near-equal, at most 510 points: the closed-form kernel stores aligned pairs) and its chunks on 512-point
        // boundaries of the batch arrays
        if (span_k > 0) {
            c_runs = 1 + n_quiet; c_span_pts = S;
            const int64_t g0 = pt_off;                           // the span starts the path
            // (counting pass: fuse_spans = fusing is possible for this batch, both alternatives are counted; fill pass: the host's
            // decision -- all fields of field work have fusable spans, or nothing is fused)
            if (!no_bases) span_counts(g0, S, is_work, fuse_spans, c_span, c_span_f, c_work_span_pts, c_unfusable);
            fused_span = c_work_span_pts;
            if (FILL && fused_span > 0) c_span = 0;          // (no chunk records for a fused span)
            if (FILL) {
                const int64_t bs = S / span_k, rem = S % span_k;
                if (lane == 0) {          // (the device's tile table keeps one tile per statistics entry: of the span's run, the first)
                    DevTile t;
                    t.field = (int32_t)field; t.start = 0; t.count = (int32_t)(bs + (rem > 0 ? 1 : 0)); t.quiet = 4; t.stat_tile = 0;
                    t.idx0 = 0; t.off0 = 0;
                    T.tiles[stat_base] = t;
                }
                const int64_t c_first = (S < TILE_POINTS - (g0 % TILE_POINTS)) ? S : TILE_POINTS - (g0 % TILE_POINTS);
                for (int64_t j = lane; j < c_span; j += 64) {
                    const int64_t done = j == 0 ? 0 : c_first + (j - 1) * TILE_POINTS;
                    const int64_t cnt = j == 0 ? c_first : ((S - done < TILE_POINTS) ? S - done : TILE_POINTS);
                    DevTile ch;
                    ch.field = (int32_t)field; ch.start = done; ch.count = (int32_t)cnt; ch.quiet = 4; ch.stat_tile = (int32_t)stat_base;
                    ch.idx0 = (int32_t)(done / per); ch.off0 = (int32_t)(done % per);
                    T.span_chunks[span_base + j] = ch;
                }
            }
        }
        c_tiles = (span_k > 0 ? 1 : 0) + n_wave + n_general + n_quiet;      // (the tile table holds one tile per statistics entry: of a span / a run, the first)
        c_wave = n_wave; c_general = n_general;
        c_stat = (span_k > 0 ? 1 : 0) + n_wave + n_general + n_quiet;
        if (dense) { c_chunks_out = c_chunks; c_chunk_pts_out = c_chunk_pts; dense_entries = true; }
    }
    // ---- which kernel reduces the field: its own workgroup (k_plan_sparse_fields) or a class of k_reduce_stats
    const int64_t ne = c_stat;
    if (is_work) { c_work = 1; c_work_wave_pts = c_wave_pts; }
    else { c_open = n_wave; cls = tiler_reduce_class(ne, tc.reduce_wg_max); }
    if (fallback && lane == 0) atomicMax(reinterpret_cast<unsigned long long *>(totals + PC_COLS + PF_FALLBACK), (unsigned long long)tc.gen);
    if (!FILL && tc.speculative && lane == 0 && (c_tiles > 1 + DEVPLAN_KEEP_TILES || c_wave > DEVPLAN_KEEP_TILES || c_general > DEVPLAN_KEEP_TILES || c_span > SPEC_SPAN_CHUNKS))
        atomicMax(reinterpret_cast<unsigned long long *>(totals + PC_COLS + PF_OVER_CAPACITY), (unsigned long long)tc.gen);

    TSTAMP(37);
    FSTAMP(2);
    if (!FILL) {
        if (lane == 0) {
            int64_t *c = counts + field;
            c[(int64_t)PC_TILES * n] = c_tiles; c[(int64_t)PC_WAVE * n] = c_wave; c[(int64_t)PC_GENERAL * n] = c_general; c[(int64_t)PC_STAT * n] = c_stat;
            c[(int64_t)PC_WORK * n] = c_work; c[(int64_t)PC_OPEN * n] = c_open;
            if (!no_bases) { c[(int64_t)PC_SPAN * n] = c_span; c[(int64_t)PC_WORK_SPAN_PTS * n] = c_work_span_pts; c[(int64_t)PC_SPAN_F * n] = c_span_f; c[(int64_t)PC_UNFUSABLE * n] = c_unfusable; }
            c[(int64_t)PC_CLS0 * n] = (!is_work && cls == 0) ? 1 : 0; c[(int64_t)PC_CLS1 * n] = (!is_work && cls == 1) ? 1 : 0;
            c[(int64_t)PC_CLS2 * n] = (!is_work && cls == 2) ? 1 : 0; c[(int64_t)PC_CLS3 * n] = (!is_work && cls == 3) ? 1 : 0;
            c[(int64_t)PC_RUNS * n] = c_runs; c[(int64_t)PC_SPAN_PTS * n] = c_span_pts; c[(int64_t)PC_WAVE_PTS * n] = c_wave_pts;
            c[(int64_t)PC_WORK_WAVE_PTS * n] = c_work_wave_pts; c[(int64_t)PC_WAVE_INSIDE * n] = c_wave_inside;
            c[(int64_t)PC_CHUNKS * n] = c_chunks_out; c[(int64_t)PC_CHUNK_PTS * n] = c_chunk_pts_out;
        }
        TSTAMP(38);
        return;
    }

    // ---- fill pass: the field's descriptor and primitives at their final places, the statistics entries, the work lists
    const bool lds_pack = STAGE && prim_count <= TW_LDS_PRIMS;        // the pack's copies of the primitives come out of LDS
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&T.fields[field]);
        if (lane < NFW) dst[lane] = fw;
        const unsigned long long *ps = reinterpret_cast<const unsigned long long *>(prims);
        unsigned long long *pd = reinterpret_cast<unsigned long long *>(T.prims + prim_base);
        if (lds_pack) wave_sync();                           // (a field cut again above has read its primitives from there)
#pragma unroll
        for (int j = 0; j < PF_ROUNDS; ++j) {
            const int k = lane + 64 * j;
            if (k < nwords_p) { pd[k] = pw[j]; if (lds_pack) L.prim_words[k] = pw[j]; }
        }
        for (int k = lane + 64 * PF_ROUNDS; k < nwords_p; k += 64) pd[k] = ps[k];
        if (lds_pack) wave_sync();
    }
    FSTAMP(3);
    // entries in path order: the span's run, then the wave tiles / general tiles
    if (span_k > 0 && lane == 0) { T.stat_ids[stat_base] = (int32_t)stat_base; T.stat_run[stat_base] = S; }
    if (!dense_entries) {
        const int64_t e0 = stat_base + (span_k > 0 ? 1 : 0), nt = n_wave + n_general;
        for (int64_t j = lane; j < nt; j += 64) { T.stat_ids[e0 + j] = (int32_t)(e0 + j); T.stat_run[e0 + j] = 0; }
    }
    if (lane == 0) {
        T.stat_first[field] = stat_base;
        if (field == n - 1) T.stat_first[n] = stat_base + c_stat;
        if (is_work) {
            DevFieldWork w;
            memset(&w, 0, sizeof w);
            w.field = (int32_t)field; w.n_tiles = (int32_t)n_wave; w.w_first = (int32_t)wave_base; w.e_first = (int32_t)stat_base; w.n_entries = (int32_t)ne;
            w.fused_span = (int32_t)fused_span;
            T.field_work[base_of(PC_WORK)] = w;              // (every such field has at most four tiles: class 0 is the only class in use)
            DevFieldPack &P = T.field_packs[base_of(PC_WORK)];
            P.work = w;
            P.span_points = fused_span;
            for (int k = 0; k < 6; ++k) P._pad0[k] = 0;
            P._pad1 = 0.0;
            for (int k = 0; k < 4; ++k) P._pad2[k] = 0.0;
        } else {
            int64_t cls_first = 0;
            for (int k = 0; k < cls; ++k) cls_first += totals[PC_CLS0 + k];
            T.red_paths[cls_first + base_of(PC_CLS0 + cls)] = (int32_t)field;
        }
    }
    {
        // connector segments (MLP:1313-1355): approach rows [0, n), departure rows [n, 2n) -- out of the lanes' words of fcpp_field_info
        constexpr int W_AF = (int)(offsetof(fcpp_field_info, approach_from) / 8), W_DF = (int)(offsetof(fcpp_field_info, departure_from) / 8);
        constexpr int W_ST = (int)(offsetof(fcpp_field_info, status) / 8), W_SK = (int)(offsetof(fcpp_field_info, start_kept) / 8), W_EK = (int)(offsetof(fcpp_field_info, end_kept) / 8);
        static_assert(offsetof(fcpp_field_info, approach_to) == offsetof(fcpp_field_info, approach_from) + 16 &&
                      offsetof(fcpp_field_info, departure_to) == offsetof(fcpp_field_info, departure_from) + 16, "four doubles each");
        auto i32_of = [&](int w, bool hi) -> int32_t { const unsigned long long v = __shfl(iw, w); return (int32_t)(hi ? (v >> 32) : (v & 0xffffffffull)); };
        const int32_t status = i32_of(W_ST, offsetof(fcpp_field_info, status) % 8 != 0);
        const int32_t start_kept = i32_of(W_SK, offsetof(fcpp_field_info, start_kept) % 8 != 0), end_kept = i32_of(W_EK, offsetof(fcpp_field_info, end_kept) % 8 != 0);
        const bool okf = status == FCPP_OK;
        // lanes 0..3: the approach's four doubles, lanes 4..7: the departure's
        const unsigned long long cw = __shfl(iw, lane < 4 ? W_AF + lane : W_DF + (lane & 3));
        unsigned long long *sg = reinterpret_cast<unsigned long long *>(T.seg);
        if (lane < 4) sg[field * 4 + lane] = cw;
        else if (lane < 8) sg[(n + field) * 4 + (lane - 4)] = cw;
        if (lane == 0) { T.seg_mask[field] = okf && start_kept; T.seg_mask[n + field] = okf && end_kept; }
    }
    FSTAMP(4);
    if (!is_work) {
        const int64_t ob = base_of(PC_OPEN);
        for (int64_t j = lane; j < n_wave; j += 64) T.open_wave_ids[ob + j] = (int32_t)(wave_base + j);
    } else {
        // the field's pack: its tiles (lane t holds tile t: at most four, all kept by the counting pass), its descriptor as written to
        // the field table, and per tile a copy of its primitives
        DevFieldPack &P = T.field_packs[base_of(PC_WORK)];
        if (lane < FIELD_WORK_TILES) P.tile[lane] = my_wt;               // (lanes without a tile hold zeros)
        {
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(&P.field);
            if (lane < NFW) dst[lane] = fw;
        }
        constexpr int PW = PWD, TW = PACK_TILE_PRIMS * PW;     // 8-byte words per primitive / per tile's copy
        const unsigned long long *psrc = reinterpret_cast<const unsigned long long *>(prims);
        unsigned long long *pdst = reinterpret_cast<unsigned long long *>(&P.prims[0][0]);
        for (int t = 0; t < FIELD_WORK_TILES; ++t) {
            const int p0 = __shfl(my_p0_rel, t), np = __shfl(my_np, t);
            for (int k = lane; k < TW; k += 64)
                pdst[t * TW + k] = (k / PW < np) ? (lds_pack ? L.prim_words[p0 * PW + k] : psrc[(int64_t)p0 * PW + k]) : 0ull;
        }
    }
    FSTAMP(5);
    // ---- what the host path computes with three more launches after its copy (k_field_junctions, k_run_consts, k_work_totals), per field:
    // the junction after a U-turn, the closed-form statistics of the field's span in its slot (zeros in the slots of its tiles), and for a
    // field of field work the sum of its runs' statistics
    {
        double2 junc = make_double2(0.0, 0.0);
        if (n_total > 0 && F.P >= 2 && F.n_line >= 2 && F.n_turn >= 1) junc.x = line_start_curvature(F, cst, 1, junc.y);
        TilePartial tp;
        memset(&tp, 0, sizeof tp);
        if (span_k > 0) {
            DevTile tl;
            tl.field = (int32_t)field; tl.start = 0; tl.count = 0; tl.quiet = 4; tl.stat_tile = 0; tl.idx0 = 0; tl.off0 = 0;
            FieldStatView fv;
            fv.n_line = F.n_line; fv.n_turn = F.n_turn; fv.reverse_order = F.reverse_order; fv.line_step = F.line_step; fv.n_main = F.n_main; fv.junc = junc;
            const DevRun run = { (int32_t)stat_base, 0, S };
            tp = quiet_run_partial(run, tl, fv, T.prims, cst);
            tp.n_viol = tp.n_outside = tp.n_in_obstacle = tp.n_adjusted = 0;
        }
        FSTAMP(6);
        static_assert(sizeof(TilePartial) == 13 * 8, "thirteen 8-byte components");
        // slots: entry 0 = the span's (when there is one), the others zero
        unsigned long long *slots = reinterpret_cast<unsigned long long *>(T.partial + stat_base);
        const unsigned long long *tpw = reinterpret_cast<const unsigned long long *>(&tp);
        const int64_t nwords = dense_entries ? 13 : c_stat * 13;          // (dense: the span's slot; the dense block wrote the others)
        for (int64_t k = lane; k < nwords; k += 64) {
            unsigned long long v = 0;
            if (span_k > 0 && k < 13) {
#pragma unroll
                for (int q = 0; q < 13; ++q) if (k == q) v = tpw[q];
            }
            slots[k] = v;
        }
        if (lane == 0) {
            T.field_junc[field] = junc;
            if (is_work) {
                TilePartial t;
                memset(&t, 0, sizeof t);
                if (span_k > 0) {
                    t.main_len += tp.main_len; t.main_time_pre += tp.main_time_pre; t.main_time += tp.main_time;
                    t.head_len += tp.head_len; t.head_time_pre += tp.head_time_pre; t.head_time += tp.head_time;
                    t.max_kappa = fmax(t.max_kappa, tp.max_kappa); t.max_alat = fmax(t.max_alat, tp.max_alat); t.max_jump = fmax(t.max_jump, tp.max_jump);
                    t.n_viol += tp.n_viol; t.n_adjusted += tp.n_adjusted;
                }
                T.work_totals[base_of(PC_WORK)] = t;
            }
        }
        // the field's fcpp_field_info stays with the batch (fcpp_batch_info copies it back when asked)
        static_assert(sizeof(fcpp_field_info) % 8 == 0, "copied as 8-byte words");
        unsigned long long *id = reinterpret_cast<unsigned long long *>(&T.info[field]);
        if (lane < NIW) id[lane] = iw;
    }
    FSTAMP(7);
}

template <bool FILL, bool STAGE, bool DENSE = false>
__global__ __launch_bounds__(64 * TW_WAVES, 4) void k_tile_fields(int64_t n, DevTileConsts tc, DevConst cst, const DevField *__restrict__ ftmp, const DevPrim *__restrict__ ptmp,
                                                              fcpp_field_info *__restrict__ info, int64_t *__restrict__ counts,
                                                              const int64_t *__restrict__ bases, int64_t *__restrict__ totals,
                                                              DevTile *__restrict__ keep_tiles, DevWaveTile *__restrict__ keep_wtiles, DevPlanTables T)
{
    tile_fields_body<FILL, STAGE, DENSE>(n, tc, cst, ftmp, ptmp, info, counts, bases, totals, keep_tiles, keep_wtiles, T);
}
// (Measured and not kept: the fill pass of a large batch held to six wavefronts per SIMD -- 85 registers, 225 spilled, the primitives not staged:
// cfg5 390 instead of 235 us.)

__global__ void k_debug_math(int fn, int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ o0, double *__restrict__ o1)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn == 0) { double s, c; fc_sincos(a[i], s, c); o0[i] = s; o1[i] = c; }
    else if (fn == 4) { double s, c; fc_sincos_cr(a[i], s, c); o0[i] = s; o1[i] = c; }
    else if (fn == 5) o0[i] = fc_atan2_cr(a[i], b[i]);
    else if (fn == 1) o0[i] = atan2_fd(a[i], b[i]);
    else if (fn == 2) o0[i] = fc_acos(a[i]);
    else o0[i] = fc_hypot(a[i], b[i]);
}

}  // namespace

int launch_devplan_count(hipStream_t st, int64_t n, const PlanConsts &pc, const DevTileConsts &tc, const DevPlanScratch &s, const fcpp_field *fields,
                         int64_t n_polys, int check_obstacles, int64_t *totals_host, hipStream_t side, hipEvent_t *ev, int n_ev)
{
    if (n <= 0) return 0;
    // (sixteen lanes per field; FCPP_PLAN_SERIAL=1 -- read once -- keeps the one-thread-per-field kernel: the A/B and the checker of the two)
    // Sixteen lanes per field cut the latency of a plan (23 instead of 35 us for 4096 fields) and, four fields per wavefront, its stores are
    // denser (cfg5's 65 536 fields: 170 us against 230 for the one-thread kernel); FCPP_PLAN_SERIAL=1 keeps the latter (the A/B, the checker)
    static const bool plan_serial = getenv("FCPP_PLAN_SERIAL") != nullptr;
    static const bool one_stream = !(getenv("FCPP_COUNT_CHUNKS") != nullptr && atoi(getenv("FCPP_COUNT_CHUNKS")) >= 2);      // (the chunks: an experiment, see below)
    // The counting pass goes without the fields' point offsets; what depends on them (span_counts) is derived by the scan that follows the
    // pass: small batches ONE scan of one launch (k_scan_small), large ones the scan of the points (its apply kernel derives) and then the
    // scan of the other columns.  (Rounds 4-5a scanned the points of a large batch BEFORE its pass.)
    const bool one_scan = (n + 1023) / 1024 <= devplan_small_blocks();
    DevTileConsts tcc = tc;
    tcc.no_bases = 1;
    auto plan = [&](hipStream_t q, int64_t f0, int64_t f1) {
        if (plan_serial) {
            if (f0 == 0)
                hipLaunchKernelGGL(k_plan_fields, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, q, n, pc, fields, s.info, s.fields_tmp, s.prims_tmp, s.counts, s.totals,
                                   n_polys, check_obstacles, 0, tc.gen);
        } else
            hipLaunchKernelGGL(k_plan_fields16, dim3((unsigned)((f1 - f0 + 3) / 4)), dim3(64), 0, q, n, pc, fields, s.info, s.fields_tmp, s.prims_tmp, s.counts, s.totals,
                               n_polys, check_obstacles, tc.gen, f0, f1);
    };
    // (more fields than one round of wavefronts takes -- 4 per SIMD x 1024 SIMDs with the primitives staged in LDS: they are not staged, five
    // wavefronts per SIMD instead of four; cfg5's 65 536 fields: plan + count 1.30 -> 1.07 ms)
    auto count = [&](hipStream_t q, int64_t f0, int64_t f1) {
        tcc.f0 = f0; tcc.f1 = f1;
        const unsigned grid = (unsigned)((f1 - f0 + TW_WAVES - 1) / TW_WAVES);
        if (tc.dense)
            hipLaunchKernelGGL((k_tile_fields<false, true, true>), dim3(grid), dim3(64 * TW_WAVES), 0, q, n, tcc, DevConst(), s.fields_tmp, s.prims_tmp,
                               s.info, s.counts, s.bases, s.totals, s.keep_tiles, s.keep_wtiles, DevPlanTables());
        else if (n <= 4096)
            hipLaunchKernelGGL((k_tile_fields<false, true>), dim3(grid), dim3(64 * TW_WAVES), 0, q, n, tcc, DevConst(), s.fields_tmp, s.prims_tmp,
                               s.info, s.counts, s.bases, s.totals, s.keep_tiles, s.keep_wtiles, DevPlanTables());
        else
            hipLaunchKernelGGL((k_tile_fields<false, false>), dim3(grid), dim3(64 * TW_WAVES), 0, q, n, tcc, DevConst(), s.fields_tmp, s.prims_tmp,
                               s.info, s.counts, s.bases, s.totals, s.keep_tiles, s.keep_wtiles, DevPlanTables());
    };
    // Large batches in CHUNKS on two streams (FCPP_COUNT_CHUNKS=2..4; measured and NOT the default): the planner's first load is the transfer
    // of the field records (128 bytes a field over PCIe: the kernel runs at the link's 50 GB/s), the counting pass of a chunk needs nothing but
    // its own fields' plans -- beside the planner's next chunk instead of behind its last.  It loses: the planner reaches the link's rate only
    // with the whole chip's wavefronts waiting on loads, and beside the counting pass it has half of them -- cfg5, four chunks: planner
    // 60 + 66 + 76 + 85 us instead of 171, counting pass 106 + 86 + 76 + 71 instead of 249, the scans begin at 460 us instead of 421.
    int n_chunks = 1;
    if (!one_scan && !plan_serial && !one_stream && side && side != st && ev && n_ev >= 2) {
        n_chunks = n_ev - 1 < 4 ? n_ev - 1 : 4;
        if (const char *e = getenv("FCPP_COUNT_CHUNKS")) { const int c = atoi(e); if (c >= 2 && c <= n_ev - 1) n_chunks = c; }
    }
    if (tc.dense) {
        // dense sampling: the chunks of every quiet run lie on 512-point boundaries of the batch arrays -- the pass needs the point offsets
        plan(st, 0, n);
        int rc0 = launch_scan(st, n, PC_POINTS, PC_PRIMS + 1, s, totals_host, 0);
        if (rc0) return rc0;
        tcc.no_bases = 0;
        count(st, 0, n);
        rc0 = launch_scan(st, n, PC_TILES, PC_COLS, s, totals_host, 1, 0, 0, 0, tc.gen);
        if (rc0) return rc0;
        const hipError_t e0 = hipGetLastError();
        return e0 == hipSuccess ? 0 : (int)e0;
    }
    if (n_chunks <= 1) { plan(st, 0, n); count(st, 0, n); }
    else {
        const int64_t per = (((n + n_chunks - 1) / n_chunks) + 3) / 4 * 4;      // (whole wavefronts of the planner, whole workgroups of the pass)
        int used = 0;
        for (int64_t f0 = 0; f0 < n; f0 += per, ++used) {
            const int64_t f1 = f0 + per < n ? f0 + per : n;
            plan(st, f0, f1);
            hipError_t e = hipEventRecord(ev[used], st);
            if (e == hipSuccess) e = hipStreamWaitEvent(side, ev[used], 0);
            if (e != hipSuccess) return (int)e;
            count(side, f0, f1);
        }
        hipError_t e = hipEventRecord(ev[n_ev - 1], side);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ev[n_ev - 1], 0);
        if (e != hipSuccess) return (int)e;
    }
    int rc = 0;
    if (one_scan) {
        rc = launch_scan(st, n, PC_POINTS, PC_COLS, s, totals_host, 1, 1, tc.fuse_spans, tc.speculative ? tc.gen : 0, tc.gen);
        if (rc) return rc;
        const hipError_t e1 = hipGetLastError();
        return e1 == hipSuccess ? 0 : (int)e1;
    }
    rc = launch_scan(st, n, PC_POINTS, PC_PRIMS + 1, s, totals_host, 0, 1, tc.fuse_spans);
    if (rc) return rc;
    rc = launch_scan(st, n, PC_TILES, PC_COLS, s, totals_host, 1, 0, 0, 0, tc.gen);
    if (rc) return rc;
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

#ifdef FCPP_DIAG_TILE
extern "C" __attribute__((visibility("default"))) int fcpp_diag_tile_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_stamps), sizeof g_tile_stamps);
}
extern "C" __attribute__((visibility("default"))) int fcpp_diag_plan_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_plan_stamps), sizeof g_plan_stamps);
}
extern "C" __attribute__((visibility("default"))) int fcpp_diag_fill_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fill_stamps), sizeof g_fill_stamps);
}
#endif

int launch_devplan_points(hipStream_t st, int64_t n, const PlanConsts &pc, const DevPlanScratch &s, const fcpp_field *fields)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_plan_fields, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, pc, fields, s.info, s.fields_tmp, s.prims_tmp, s.counts, s.totals,
                       (int64_t)0, 0, 1, (int64_t)0);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_devplan_fill(hipStream_t st, int64_t n, const DevTileConsts &tc, const DevConst &cst, const DevPlanScratch &s, const DevPlanTables &t)
{
    if (n <= 0) return 0;
    if (tc.dense)
        hipLaunchKernelGGL((k_tile_fields<true, true, true>), dim3((unsigned)((n + TW_WAVES - 1) / TW_WAVES)), dim3(64 * TW_WAVES), 0, st, n, tc, cst, s.fields_tmp, s.prims_tmp,
                           s.info, s.counts, s.bases, s.totals, s.keep_tiles, s.keep_wtiles, t);
    else
        hipLaunchKernelGGL((k_tile_fields<true, true>), dim3((unsigned)((n + TW_WAVES - 1) / TW_WAVES)), dim3(64 * TW_WAVES), 0, st, n, tc, cst, s.fields_tmp, s.prims_tmp,
                           s.info, s.counts, s.bases, s.totals, s.keep_tiles, s.keep_wtiles, t);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_debug_math(hipStream_t st, int fn, int64_t n, const double *a, const double *b, double *out0, double *out1)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fn, n, a, b, out0, out1);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp
