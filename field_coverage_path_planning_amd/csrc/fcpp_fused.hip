// fcpp_fused.hip -- pipeline B: the whole hot path in ONE pass over HBM (36 B written per point, nothing read
// back): generate -> curvature -> geofence -> clamp -> forward/backward sweeps -> a_lat check -> metrics.
//
// Work decomposition (gfx950: 64-wide waves, 256 CUs): one 256-thread workgroup per 2048-point tile of one
// field's path; thread t owns the 8 CONSECUTIVE points 8t..8t+7, so
//   * the 3-point curvature stencil and the segment lengths live in registers; only the two end
//     neighbours of a thread come from the adjacent lane (shuffle) or, at wave edges, from 64 bytes of LDS;
//   * the min-plus scans of the sweeps run over registers: 8 serial steps per thread, one 6-step wave scan
//     of the per-thread maps, 4 wave aggregates through LDS;
//   * results are transposed through a 4.5 KB per-wave LDS buffer so that every global store instruction
//     writes 512 contiguous bytes per wave (SoA arrays, coalesced).
//
// Turn geometry comes from two small per-batch TEMPLATES (k_build_templates, run once at batch creation): every
// U-turn of a batch is a translate/mirror of one sampled shape and every corner turn a quadrant rotation of
// another, so the kernel never evaluates sincos or Fresnel series: a turn point is one 16-byte load and two adds,
// bit-identical to the direct formula.
//
// No workgroup ever waits for another one: what the sweeps need from outside the tile is RECOMPUTED.
// A constraint at point j can only bind at point i while u0_j + 2a*dist(i,j) < u_cap = (v_max/3.6)^2, i.e.
// within u_cap/(2a) metres (5.8 m for the default vehicle).  If that stretch next to the tile is one straight
// primitive the carried value is its nominal u0 (closed form); otherwise wave 0 (wave 3) re-generates 64-point
// chunks before (after) the tile until the accumulated 2a*distance exceeds u_cap or the path ends.  The
// recomputation uses the same arithmetic as the owning tile, so results do not depend on the tiling.
#include <algorithm>
#include "fcpp_quiet_fn.h"

namespace fcpp {

// the fused kernel runs ONE wavefront per 512-point tile: no workgroup barrier anywhere, waves never wait for each other
static constexpr int FBLOCK = 64;
static constexpr int FIPT = TILE_POINTS / FBLOCK;   // 8 consecutive points per lane
static constexpr int FNWAVE = 1;

struct HaloInfo {
    double px, py;       // the neighbouring path point (index s-1 or s+count)
    double carry;        // value the forward (backward) sweep carries into the tile; +inf if none
    double kappa, v0, u0;
    uint32_t fs;
    int valid;
};

struct RedSharedF { double d[FNWAVE][9]; long long i[FNWAVE][4]; };

static constexpr int TR_WORDS = 64 * FIPT + 64;   // padded per-wave transposition buffer (doubles)

struct FieldWords { uint32_t w[sizeof(DevField) / 4]; };
static_assert(sizeof(DevField) % 4 == 0 && sizeof(DevField) / 4 <= 2 * FBLOCK, "DevField staging");

static constexpr int FPRIM_CAP = 48;     // headland primitives of one field staged in LDS (a field has <= 40)
static_assert(sizeof(DevPrim) % 4 == 0, "DevPrim staging");

struct FusedShared {
    FieldWords fw;       // the tile's field descriptor, staged by ONE coalesced load
    uint32_t prims[FPRIM_CAP * sizeof(DevPrim) / 4];     // the field's headland primitives (tiles that touch layer 2)
    NomTable nom;
    double cd[6];        // a_lat, a_lon, sf, geofence_tol, u_cap, inv_sf36
    // double number q of the staged descriptor, as a wave-uniform (scalar) value
    __device__ __forceinline__ double fw_d(int q) const
    {
        const uint32_t lo = __builtin_amdgcn_readfirstlane(fw.w[2 * q]), hi = __builtin_amdgcn_readfirstlane(fw.w[2 * q + 1]);
        return __hiloint2double((int)hi, (int)lo);
    }
    HaloInfo back, fwd;
    Agg wf[FNWAVE], wb[FNWAVE];
    double ev[FNWAVE], ek[FNWAVE];                                // last item of each wave: final v, kappa
    uint32_t efs[FNWAVE];                                         // ... and its segment word
    double tr[FNWAVE][TR_WORDS];
    RedSharedF R;
};

// random access (halo recomputation only): point i of the field's path
__device__ __noinline__ GenOut gen_point_tmpl(const DevField *f, const PrimTable prims, const DevConst *cst, int64_t i)
{
    GenOut o;
    o.v = 0;
    if (i < f->gen_main) {
        const int64_t per = (int64_t)f->n_line + f->n_turn;
        const int64_t idx = i / per;
        eval_main(*f, *cst, (int)idx, (int)(i - idx * per), o.x, o.y, o.fs);
    } else {
        const DevPrim &p = prims[find_prim(*f, prims, i)];
        eval_prim(p, *cst, (int)(i - p.start), o.x, o.y);
        o.fs = p.fs;
    }
    return o;
}

// One wave computes what the sweeps carry across the tile edge (see the header comment).
template <bool BACK>
__device__ void halo_wave(const DevField &f, const DevField *fg, const PrimTable prims, const DevConst &cst, const NomTable &nom,
                          const DevTile &tl, int64_t edge, HaloInfo *out)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = f.n_total;
    const double two_a = 2 * cst.a_lon;
    if (BACK ? (edge <= 0) : (edge >= n)) {
        if (lane == 0) { out->valid = 0; out->carry = FCPP_INF; out->px = out->py = 0; out->kappa = out->v0 = out->u0 = 0; out->fs = 0; }
        return;
    }
    // Fast case: the point next to the edge lies on a straight primitive that extends at least u_cap / (2a) metres
    // away from the tile.  Every point on that stretch has curvature 0 and the same u0, and nothing farther away can
    // bind, so the carried value is that u0 and no point needs generating.
    {
        const int64_t nb = BACK ? edge - 1 : edge;      // the neighbouring point
        double ax = 0, ay = 0, bx = 0, by = 0, sx = 0, sy = 0;
        int pos = -1, np = 0;
        uint32_t fw = 0;
        bool rot = false;
        if (nb < f.gen_main) {
            if (tl.start < f.gen_main) {
                const int per = f.n_line + f.n_turn;
                int off = tl.off0 + (int)(nb - tl.start), idx = tl.idx0;
                if (off >= 0) {
                    if (off >= per) { const int q = off / per; off -= q * per; idx += q; }
                    if (off < f.n_line) {
                        const int pi = f.reverse_order ? (f.P - 1 - idx) : idx;
                        const bool go_left = f.start_from_right ? ((idx & 1) == 0) : ((idx & 1) == 1);
                        ax = go_left ? f.lex : f.lsx; bx = go_left ? f.lsx : f.lex; sx = go_left ? -f.line_step : f.line_step;
                        ay = by = f.min_y + (double)pi * f.W; sy = 0.0;
                        pos = off; np = f.n_line; rot = f.rotated != 0;
                        fw = FCPP_KIND_SWATH | ((uint32_t)pi << FCPP_INDEX_SHIFT);
                    }
                }
            }
        } else {
            const DevPrim &p = prims[find_prim(f, prims, nb)];
            if (p.kind == PRIM_LINSPACE) {
                ax = p.a[0]; bx = p.a[2]; sx = p.a[4]; ay = p.a[1]; by = p.a[3]; sy = p.a[5];
                pos = (int)(nb - p.start); np = p.n; fw = p.fs;
            }
        }
        if (pos >= 0) {
            const double step_len = sqrt(sx * sx + sy * sy);
            if (step_len >= 1e-6) {
                // nothing farther than u_nominal / (2a) can lower the speeds ON this straight below its nominal value
                const double msn = nom_ms(nom, fw);
                const double needd = (msn * msn) / (two_a * step_len) + 3.0;
                // the neighbour's own curvature stencil must lie on the primitive too: it may not be the sample that
                // faces the tile's side of the primitive's end (its kappa would see the next primitive)
                const bool stencil_ok = BACK ? (pos <= np - 2) : (pos >= 1);
                const bool ok = stencil_ok && (BACK ? ((double)pos - needd >= 0.0) : ((double)pos + needd <= (double)(np - 1)));
                if (ok) {
                    if (lane == 0) {
                        const double ms = msn;
                        double px = (double)pos * sx + ax, py = (double)pos * sy + ay;
                        if (pos == np - 1) { px = bx; py = by; }
                        if (rot) rotate_back(f, px, py);
                        out->valid = 1; out->px = px; out->py = py; out->kappa = 0.0; out->fs = fw;
                        out->v0 = nom_v(nom, fw); out->u0 = ms * ms; out->carry = ms * ms;
                    }
                    return;
                }
            }
        }
    }
    Agg total = { FCPP_INF, 0.0 };
    bool first = true;
    int64_t b = BACK ? edge - 64 : edge;
    for (;;) {
        const int64_t i = b + lane;
        const bool act = i >= 0 && i < n;
        GenOut g; g.x = g.y = g.v = 0; g.fs = 0;
        if (act) g = gen_point_tmpl(fg, prims, &cst, i);
        double xm = __shfl_up(g.x, 1), ym = __shfl_up(g.y, 1), xp = __shfl_down(g.x, 1), yp = __shfl_down(g.y, 1);
        const int64_t ei = lane == 0 ? i - 1 : i + 1;
        if ((lane == 0 || lane == 63) && ei >= 0 && ei < n) {   // the chunk's two outer neighbours
            const GenOut e = gen_point_tmpl(fg, prims, &cst, ei);
            if (lane == 0) { xm = e.x; ym = e.y; } else { xp = e.x; yp = e.y; }
        }
        double kappa = 0, dprev = 0, dnext = 0;
        const double dx1 = g.x - xm, dy1 = g.y - ym, dx2 = xp - g.x, dy2 = yp - g.y;
        if (act && i > 0) dprev = seg_len(dx1, dy1);
        if (act && i < n - 1) dnext = seg_len(dx2, dy2);
        if (act && i > 0 && i < n - 1) kappa = curv_chords(dx1, dy1, dprev, dx2, dy2, dnext);
        bool cl;
        const double v0 = clamped_speed(nom_v(nom, g.fs), kappa, cst, cl);
        const double ms = cl ? v0 / 3.6 : nom_ms(nom, g.fs);
        Agg me;
        me.c = act ? ms * ms : FCPP_INF;
        if (BACK) me.w = !act ? 0.0 : ((i == 0 || dprev < 1e-6) ? FCPP_INF : two_a * dprev);
        else      me.w = !act ? 0.0 : ((i == n - 1 || dnext < 1e-6) ? FCPP_INF : two_a * dnext);
        Agg inc = me;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            if (BACK) {
                Agg p = { __shfl_up(inc.c, o), __shfl_up(inc.w, o) };
                if (lane >= o) inc = combine_after(p, inc);
            } else {
                Agg p = { __shfl_down(inc.c, o), __shfl_down(inc.w, o) };
                if (lane + o < 64) inc = combine_after(p, inc);
            }
        }
        const int src = BACK ? 63 : 0;
        Agg chunk = { __shfl(inc.c, src), __shfl(inc.w, src) };
        total = first ? chunk : combine_after(chunk, total);   // the farther chunk acts first
        if (first && lane == src) {
            out->valid = 1; out->px = g.x; out->py = g.y; out->kappa = kappa; out->v0 = v0; out->fs = g.fs; out->u0 = me.c;
        }
        first = false;
        const bool done = (total.w >= cst.u_cap) || (BACK ? (b <= 0) : (b + 64 >= n));
        if (done) break;
        b += BACK ? -64 : 64;
    }
    if (lane == 0) out->carry = total.c;
}

// per field, once per batch: the junction after a U-turn (all lines of a field are mirror images of each other)
__global__ void k_field_junctions(int64_t n_fields, const DevField *__restrict__ fields, DevConst cst, double2 *__restrict__ junc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_fields) return;
    const DevField &q = fields[i];
    double2 o = make_double2(0.0, 0.0);
    if (q.n_total > 0 && q.P >= 2 && q.n_line >= 2 && q.n_turn >= 1) o.x = line_start_curvature(q, cst, 1, o.y);
    junc[i] = o;
}

// the quiet path: one tile per wavefront, four per workgroup; pure HBM streaming at full occupancy
template <int KINDS, bool SCALAR_DESC, bool STAGED, bool OBS, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void k_plan_quiet(const DevTile *__restrict__ chunks, const DevField *__restrict__ fields,
                                                      const DevPrim *__restrict__ prims, DevConst cst, DevObstacles obs,
                                                      double *__restrict__ xo,
                                                      double *__restrict__ yo, double *__restrict__ ko,
                                                      double *__restrict__ vo, uint32_t *__restrict__ fso,
                                                      TilePartial *__restrict__ partial, int64_t n_chunks)
{
    __shared__ double obs_lds[WPB][OBS ? 2 * OBS_LDS_VERTS : 1];
    __shared__ double tmpl_lds[WPB][STAGED ? 3 * TMPL_LDS : 1];
    // the wave index as a scalar: the chunk and field descriptors are then fetched by scalar loads and live in scalar registers
    // (as per-lane copies of the same values they cost ~40 vector registers, i.e. one wave per SIMD of occupancy)
    const int wave = SCALAR_DESC ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const int64_t slot = (int64_t)blockIdx.x * WPB + wave;   // one chunk per wavefront
    if (slot >= n_chunks) return;
    if (STAGED) stage_turn_template(cst, tmpl_lds[wave]);  // (its loads are in flight while the descriptors arrive)
    const DevTile tl = chunks[slot];
    quiet_tile<KINDS, STAGED, OBS>(tl, &fields[tl.field], prims, cst, obs, obs_lds[wave], tmpl_lds[wave], xo, yo, ko, vo, fso,
                      reinterpret_cast<unsigned long long *>(&partial[tl.stat_tile].n_outside),
                      reinterpret_cast<unsigned long long *>(&partial[tl.stat_tile].n_in_obstacle));
}

// Diagnostic build only (-DFCPP_DIAG_STAMPS, never shipped): phase time stamps of wave 1 replace the tile's metrics.
#ifdef FCPP_DIAG_STAMPS
#define FCPP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FCPP_STAMP(i) do { } while (0)
#endif

// A kernel-argument struct arrives as ONE wide scalar-register tuple, and under register pressure the compiler spills and
// reloads the whole tuple around every use of one member.  The few constants the kernel needs therefore take the same
// route as the field descriptor: through LDS, each one becoming an independent scalar value.
__device__ __forceinline__ double uniform_d(const double *lds)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(lds);
    return __hiloint2double((int)__builtin_amdgcn_readfirstlane(w[1]), (int)__builtin_amdgcn_readfirstlane(w[0]));
}

template <int MINW>
__global__ __launch_bounds__(FBLOCK, MINW) void k_plan_fused(const DevTile *__restrict__ tiles,
                                                            const DevField *__restrict__ fields,
                                                            const DevPrim *__restrict__ prims_g, DevConst cst_arg, DevObstacles obs,
                                                            double *__restrict__ xo, double *__restrict__ yo,
                                                            double *__restrict__ ko, double *__restrict__ vo,
                                                            uint32_t *__restrict__ fso, TilePartial *__restrict__ partial,
                                                            const int32_t *__restrict__ ids)
{
    __shared__ FusedShared S;
#ifdef FCPP_DIAG_STAMPS
    unsigned long long stamp[12];
#endif
    FCPP_STAMP(0);
    const int tile_id = ids ? ids[blockIdx.x] : (int)blockIdx.x;
    const DevTile tl = tiles[tile_id];
    const DevField *fg = &fields[tl.field];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 8) { S.nom.v[tid] = nominal_speed((uint32_t)tid, cst_arg); S.nom.ms[tid] = nominal_ms((uint32_t)tid, cst_arg); }
    if (tid == 8) {
        S.cd[0] = cst_arg.a_lat; S.cd[1] = cst_arg.a_lon; S.cd[2] = cst_arg.sf; S.cd[3] = cst_arg.geofence_tol;
        S.cd[4] = cst_arg.u_cap; S.cd[5] = cst_arg.inv_sf36;
    }
    // The field descriptor is block-uniform and used all over the kernel.  Read lazily it costs dozens of dependent
    // scalar-load round trips per wave; instead: one coalesced vector load into LDS, one barrier, then every word is
    // broadcast-read and moved to a scalar register (readfirstlane) in one go.
    for (int q = tid; q < (int)(sizeof(DevField) / 4); q += FBLOCK) S.fw.w[q] = reinterpret_cast<const uint32_t *>(fg)[q];
    __syncthreads();
    // everything but the geofence edges (fetched where they are used: 24 scalar registers less across the kernel)
    static constexpr int FW_HEAD = (int)(offsetof(DevField, ex) / 4);
    FieldWords fwl;
#pragma unroll
    for (int q = 0; q < (int)(sizeof(DevField) / 4); ++q) fwl.w[q] = q < FW_HEAD ? __builtin_amdgcn_readfirstlane(S.fw.w[q]) : 0u;
    const DevField f = __builtin_bit_cast(DevField, fwl);
    const NomTable &nom = S.nom;
    DevConst cst = cst_arg;
    cst.a_lat = uniform_d(&S.cd[0]); cst.a_lon = uniform_d(&S.cd[1]); cst.sf = uniform_d(&S.cd[2]);
    cst.geofence_tol = uniform_d(&S.cd[3]); cst.u_cap = uniform_d(&S.cd[4]); cst.inv_sf36 = uniform_d(&S.cd[5]);
    const int64_t n = f.n_total, s = tl.start;
    const int cnt = tl.count;
    const double two_a = 2 * cst.a_lon;
    // A tile that reaches layer 2 looks primitives up all the time (binary searches in the halos and at the lane starts, one
    // record per generated item): from global memory every one of those is a dependent ~1 us round trip.  The field's few
    // primitives are staged in LDS once, by one coalesced load; `prims` is indexed by the batch-wide primitive index either way.
    PrimTable prims = { prims_g, 0 };
    if (s + cnt >= f.gen_main && f.prim_count <= FPRIM_CAP) {     // (block-uniform)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(prims_g + f.prim_first);
        for (int q = tid; q < f.prim_count * (int)(sizeof(DevPrim) / 4); q += FBLOCK) S.prims[q] = src[q];
        __syncthreads();
        prims = { reinterpret_cast<const DevPrim *>(S.prims), f.prim_first };
    }

    FCPP_STAMP(10);
    halo_wave<true>(f, fg, prims, cst, nom, tl, s, &S.back);          // one wave per tile: it computes both carries itself
    FCPP_STAMP(11);
    halo_wave<false>(f, fg, prims, cst, nom, tl, s + cnt, &S.fwd);

    FCPP_STAMP(1);
    // ---- 0. where this thread's run sits relative to the path's special indices (small ints from here on) ----
    const int j0 = tid * FIPT;
    const int64_t i0 = s + j0;
    const int nvalid = min(max(cnt - j0, 0), FIPT);
    const bool at_start = (i0 == 0);                                  // item 0 is the path's first point
    const int64_t rem_end = (n - 1) - i0, rem_seam = f.n_main - i0;
    const int k_end = (rem_end >= 0 && rem_end < FIPT) ? (int)rem_end : 1000;     // item that is the path's last point
    const int k_seam = rem_seam < 0 ? -1 : (rem_seam >= FIPT ? 1000 : (int)rem_seam);  // item with index n_main (first of layer 2)
    // the same for the closed-form generator: item with index gen_main (= n_main, or 0 when layer 1 is a list of primitives)
    const int64_t rem_gen = f.gen_main - i0;
    const int k_gen = rem_gen < 0 ? -1 : (rem_gen >= FIPT ? 1000 : (int)rem_gen);
    double *buf = S.tr[wave];
    const int64_t g0 = f.pt_off + s + wave * (64 * FIPT);
    const int cw = min(max(cnt - wave * (64 * FIPT), 0), 64 * FIPT);
    // coalesced SoA store of one per-item array through the per-wave transposition buffer
    auto put = [&](double *__restrict__ dst, const double *vals) {
#pragma unroll
        for (int k = 0; k < FIPT; ++k) buf[lidx(lane * FIPT + k)] = vals[k];
        wave_sync();
#pragma unroll
        for (int m = 0; m < FIPT; ++m) {
            const int p = lane + 64 * m;
            if (p < cw) dst[g0 + p] = buf[lidx(p)];
        }
        wave_sync();
    };

    // ---- 1. generate this thread's 8 consecutive points -------------------------------------------
    // Position of the first item: layer 1 = (pass idx, offset in the pass) from the tile's host-precomputed decode,
    // layer 2 = (primitive, offset) by binary search.  Three generators:
    //   A. all 8 items in layer 1: branch-free (line sample or U-turn template sample, selected per item; a pass has
    //      at least 8 points, so the run crosses at most one pass boundary) -- the lanes of a wave do not diverge over line / turn / mixed runs;
    //   B. all 8 items on one headland straight: 8 x (cvt, mul, add);
    //   C. everything else (the seam, corners, reverse fills, the path's end): a cursor walks the run item by item.
    double X[FIPT + 2], Y[FIPT + 2];
    uint32_t fs[FIPT];
    const bool in_main0 = k_gen > 0;             // item 0 is generated from the closed form of layer 1
    const int per = f.n_line + f.n_turn;
    int c_idx = 0, c_off = 0, c_a = f.prim_first, c_r = 0;
    if (in_main0) {
        c_off = tl.off0 + j0; c_idx = tl.idx0;
        if (c_off >= per) { const int q = c_off / per; c_off -= q * per; c_idx += q; }
    } else if (nvalid > 0) {
        c_a = find_prim(f, prims, i0);
        c_r = (int)(i0 - prims[c_a].start);
    }
    bool straight = false, turn_run = false;   // uniform runs: one straight primitive / one U-turn (shortcuts further down)
    bool generated = false;
    uint32_t run_fs = 0;
    if (k_gen >= FIPT && per >= FIPT) {        // A: items 0..7 all have indices < gen_main (padding items beyond the tile's count included)
        const int idx1 = c_idx + 1;
        const int pi0 = f.reverse_order ? (f.P - 1 - c_idx) : c_idx, pi1 = f.reverse_order ? (f.P - 1 - idx1) : idx1;
        const double y0 = f.min_y + (double)pi0 * f.W, y1 = f.min_y + (double)pi1 * f.W;
        const bool gl0 = f.start_from_right ? ((c_idx & 1) == 0) : ((c_idx & 1) == 1);    // go_left of pass c_idx; the next pass is the opposite
        const bool arc = f.turn_model == FCPP_TURN_ARC;
        const double xr = arc ? f.max_x : (f.max_x - f.R), xl = arc ? f.min_x : (f.min_x + f.R);
        const int nl = f.n_line;
#pragma unroll
        for (int k = 0; k < FIPT; ++k) {
            int off = c_off + k;
            const bool nxt = off >= per;
            off = nxt ? off - per : off;
            const bool go_left = gl0 != nxt;
            const double y = nxt ? y1 : y0;
            const bool is_line = off < nl;
            // line sample (numpy.linspace: k*step + start, the last one is `stop`)
            double lx = go_left ? ((double)off * -f.line_step + f.lex) : ((double)off * f.line_step + f.lsx);
            if (off == nl - 1) lx = go_left ? f.lsx : f.lex;
            // turn sample: translate / mirror of the template (MLP:815-823 or the clothoid form)
            const double2 t = cst.tmpl_u[is_line ? 0 : off - nl];
            const bool turn_right = !go_left;
            const double tx = arc ? (turn_right ? (xr - t.x) : (xl + t.x)) : (turn_right ? (xr + t.x) : (xl - t.x));
            double px = is_line ? lx : tx, py = is_line ? y : (y + t.y);
            if (f.rotated) rotate_back(f, px, py);
            X[k + 1] = px; Y[k + 1] = py;
            fs[k] = (is_line ? (uint32_t)FCPP_KIND_SWATH : (uint32_t)FCPP_KIND_UTURN) | ((uint32_t)(nxt ? pi1 : pi0) << FCPP_INDEX_SHIFT);
        }
        generated = true;
        if (nvalid == FIPT) {
            straight = c_off + FIPT <= nl;
            turn_run = c_off >= nl && c_off + FIPT <= per;
            run_fs = (straight ? (uint32_t)FCPP_KIND_SWATH : (uint32_t)FCPP_KIND_UTURN) | ((uint32_t)pi0 << FCPP_INDEX_SHIFT);
        }
    } else if (nvalid == FIPT && !in_main0) {   // B
        const DevPrim &p = prims[c_a];
        if (p.kind == PRIM_LINSPACE && c_r + FIPT <= p.n) {
            const double ax = p.a[0], bx = p.a[2], sx = p.a[4], ay = p.a[1], by = p.a[3], sy = p.a[5];
            const int nl = p.n;
            run_fs = p.fs;
#pragma unroll
            for (int k = 0; k < FIPT; ++k) {
                const int rk = c_r + k;
                double px = (double)rk * sx + ax, py = (double)rk * sy + ay;   // numpy.linspace: k*step + start
                if (rk == nl - 1) { px = bx; py = by; }                        // ... and the last sample is `stop`
                X[k + 1] = px; Y[k + 1] = py; fs[k] = run_fs;
            }
            straight = true; generated = true;
        }
    }
    if (!generated) {           // C
        bool in_main = in_main0;
#pragma unroll 1
        for (int k = 0; k < FIPT; ++k) {      // rolled: one copy of the evaluators (unrolling costs more in registers than it saves)
            double px = 0.0, py = 0.0;
            uint32_t fw = 0;
            if (k < nvalid) {
                if (in_main) {
                    eval_main(f, cst, c_idx, c_off, px, py, fw);
                    ++c_off;
                    if (c_off == per || (c_idx == f.P - 1 && c_off == f.n_line)) {
                        c_off = 0; ++c_idx;
                        if (c_idx == f.P) { in_main = false; c_a = f.prim_first; c_r = 0; }
                    }
                } else {
                    const DevPrim &p = prims[c_a];
                    fw = p.fs;
                    eval_prim(p, cst, c_r, px, py);
                    ++c_r;
                    if (c_r == p.n) { ++c_a; c_r = 0; }
                }
            }
            // compile-time indices only (runtime-indexed register arrays would go to scratch)
#pragma unroll
            for (int q = 0; q < FIPT; ++q) if (q == k) { X[q + 1] = px; Y[q + 1] = py; fs[q] = fw; }
        }
    }
    FCPP_STAMP(2);
    // end neighbours: previous thread's last point, next thread's first point
    X[0] = __shfl_up(X[FIPT], 1); Y[0] = __shfl_up(Y[FIPT], 1);
    X[FIPT + 1] = __shfl_down(X[1], 1); Y[FIPT + 1] = __shfl_down(Y[1], 1);
    __syncthreads();            // (the halos in LDS)
    if (lane == 0) { X[0] = S.back.px; Y[0] = S.back.py; }
    if (lane == 63) { X[FIPT + 1] = S.fwd.px; Y[FIPT + 1] = S.fwd.py; }
    // a tile may hold fewer than 512 points anywhere on the path: the successor of its last point is the forward halo point
    if (nvalid > 0 && j0 + nvalid == cnt) {
#pragma unroll
        for (int q = 1; q <= FIPT; ++q) if (q == nvalid) { X[q + 1] = S.fwd.px; Y[q + 1] = S.fwd.py; }
    }

    FCPP_STAMP(3);
    // ---- 2. segment lengths, curvature; everything that needs coordinates; then the coordinates leave --------
    // d[k] = |P(item k) - P(item k-1)|; item -1 / item FIPT are the end neighbours.  Bit k of `cut` = the sweeps do
    // not propagate across segment k: skipped steps (d < 1e-6, MLP:560-561 / 576-577) and the path ends.
    // Bit k of `nocouple` = segment k lies beyond the tile's last point (both ends are padding): coupling 0, the identity.
    double d[FIPT + 1], kap[FIPT];
    // Inside a U-turn both are properties of the turn's shape, the same for every turn of the batch (tmpl_u_dk): segment k
    // lies inside a turn if item k is turn sample >= 1, the curvature stencil of item k if it is sample 1 .. n_turn - 2.
    // (Layer 1 only; what remains for the generic code below are line samples -- axis-aligned: no square root, collinear:
    // no angle -- and the junctions.)
    unsigned tmpl_d = 0, tmpl_k = 0;
    if (k_gen >= FIPT && per >= FIPT) {
#pragma unroll
        for (int k = 0; k <= FIPT; ++k) {
            int off = c_off + k;
            off = off >= per ? off - per : off;
            const int tk = off - f.n_line;
            if (tk >= 1) {
                const double2 m = cst.tmpl_u_dk[tk];
                d[k] = m.x; tmpl_d |= 1u << k;
                if (k < FIPT && tk <= f.n_turn - 2) { kap[k] = m.y; tmpl_k |= 1u << k; }
            }
        }
    }
    unsigned cut = 0, nocouple = 0;
#pragma unroll
    for (int k = 0; k <= FIPT; ++k) {
        if (!((tmpl_d >> k) & 1u)) d[k] = seg_len(X[k + 1] - X[k], Y[k + 1] - Y[k]);
        if ((d[k] < 1e-6) || (k == 0 && at_start) || (k == k_end + 1)) cut |= 1u << k;
        if (k > nvalid || nvalid == 0) nocouple |= 1u << k;
    }
    // curvature: the unrolled loop handles the cheap cases (collinear, tiny and small turning angles); the rare large
    // angles (junctions, coarse reference-sampled arcs) go through ONE copy of atan2 below, one item per lane per round
    unsigned hard = 0;
#pragma unroll
    for (int k = 0; k < FIPT; ++k) {
        if ((tmpl_k >> k) & 1u) { if (k >= nvalid) kap[k] = 0.0; continue; }
        double kk = 0.0;
        if (k < nvalid && !(k == 0 && at_start) && k != k_end) {
            bool slow;
            kk = curv_chords_fast(X[k + 1] - X[k], Y[k + 1] - Y[k], d[k], X[k + 2] - X[k + 1], Y[k + 2] - Y[k + 1], d[k + 1], slow);
            if (slow) hard |= 1u << k;
        }
        kap[k] = kk;
    }
    while (__ballot(hard != 0u)) {
        if (hard) {
            const int kh = __ffs(hard) - 1;
            hard &= hard - 1;
            double x0 = 0, y0 = 0, x1 = 0, y1 = 0, x2 = 0, y2 = 0, dsum = 1;
#pragma unroll
            for (int q = 0; q < FIPT; ++q)
                if (q == kh) { x0 = X[q]; y0 = Y[q]; x1 = X[q + 1]; y1 = Y[q + 1]; x2 = X[q + 2]; y2 = Y[q + 2]; dsum = d[q] + d[q + 1]; }
            const double dx1 = x1 - x0, dy1 = y1 - y0, dx2 = x2 - x1, dy2 = y2 - y1;
            const double kk = fabs(2 * atan2_fd(dx1 * dy2 - dy1 * dx2, dx1 * dx2 + dy1 * dy2) / dsum);
#pragma unroll
            for (int q = 0; q < FIPT; ++q) if (q == kh) kap[q] = kk;
        }
    }
    int nout = 0, nobs = 0;
    {
        // geofence; convexity: a straight run whose two end points pass it lies inside as a whole
        double ex[4], ey[4], eo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ex[e] = S.fw_d(offsetof(DevField, ex) / 8 + e); ey[e] = S.fw_d(offsetof(DevField, ey) / 8 + e); eo[e] = S.fw_d(offsetof(DevField, eo) / 8 + e);
        }
        bool run_inside = false;
        const double ntol = -cst.geofence_tol;
        if (straight) {
            bool out = false;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out = out | (ex[e] * X[1] + ey[e] * Y[1] + eo[e] < ntol) | (ex[e] * X[FIPT] + ey[e] * Y[FIPT] + eo[e] < ntol);
            run_inside = !out;
        }
        const int ob0 = f.obs_first, ob1 = f.obs_first + f.obs_count;
        if (!run_inside) {
#pragma unroll
            for (int k = 0; k < FIPT; ++k) {     // (branch-free: the tests are cheaper than the jumps around them)
                const double px = X[k + 1], py = Y[k + 1];
                bool out = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) out = out | (ex[e] * px + ey[e] * py + eo[e] < ntol);
                out = out & (k < nvalid);
                nout += out ? 1 : 0;
                fs[k] |= out ? FCPP_FLAG_OUTSIDE : 0u;
            }
        }
        if (ob1 > ob0) {     // wave-uniform: bounding box of the wave's points, then culled + LDS-staged polygon tests
            double mnx = FCPP_INF, mny = FCPP_INF, mxx = -FCPP_INF, mxy = -FCPP_INF;
#pragma unroll
            for (int k = 0; k < FIPT; ++k)
                if (k < nvalid) { mnx = fmin(mnx, X[k + 1]); mxx = fmax(mxx, X[k + 1]); mny = fmin(mny, Y[k + 1]); mxy = fmax(mxy, Y[k + 1]); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mnx = fmin(mnx, __shfl_xor(mnx, o)); mny = fmin(mny, __shfl_xor(mny, o));
                mxx = fmax(mxx, __shfl_xor(mxx, o)); mxy = fmax(mxy, __shfl_xor(mxy, o));
            }
            double ox[FIPT], oy[FIPT];
#pragma unroll
            for (int k = 0; k < FIPT; ++k) { ox[k] = X[k + 1]; oy[k] = Y[k + 1]; }
            const unsigned m = obstacle_mask<FIPT>(obs, ob0, ob1, buf, mnx, mny, mxx, mxy, ox, oy, nvalid);
#pragma unroll
            for (int k = 0; k < FIPT; ++k)
                if ((m >> k) & 1u) { ++nobs; fs[k] |= FCPP_FLAG_OBSTACLE; }
            wave_sync();
        }
    }
    FCPP_STAMP(4);
    put(xo, &X[1]);
    put(yo, &Y[1]);
    put(ko, kap);
    FCPP_STAMP(5);

    // ---- 3. curvature clamp (MLP:490-504) -> u0 = (v/3.6)^2 ---------------------------------------------------
    double c[FIPT];
    int adj = 0;
    unsigned clmask = 0;    // items slowed by the clamp
    const double ms_run = nom_ms(nom, run_fs), vn_run = nom_v(nom, run_fs);
#pragma unroll
    for (int k = 0; k < FIPT; ++k) {
        double ms = (straight || turn_run) ? ms_run : nom_ms(nom, fs[k]);
        if (kap[k] > 1e-6) {
            bool cl;
            const double vc = clamped_speed((straight || turn_run) ? vn_run : nom_v(nom, fs[k]), kap[k], cst, cl);
            if (cl) { ms = vc / 3.6; clmask |= 1u << k; ++adj; }
        }
        c[k] = (k < nvalid) ? ms * ms : FCPP_INF;
    }

    // ---- 4. forward / backward sweeps as min-plus scans over registers ----------------------------
    // (lanes / items beyond the tile's last point are identities: c = +inf, coupling 0)
    auto wk = [&](int k) -> double {   // coupling across segment k
        return ((nocouple >> k) & 1u) ? 0.0 : (((cut >> k) & 1u) ? FCPP_INF : two_a * d[k]);
    };
    Agg fa = { FCPP_INF, 0.0 }, ba = { FCPP_INF, 0.0 };
#pragma unroll
    for (int k = 0; k < FIPT; ++k) { const double w = wk(k); fa.c = fmin(c[k], fa.c + w); fa.w += w; }
#pragma unroll
    for (int k = FIPT - 1; k >= 0; --k) { const double w = wk(k + 1); ba.c = fmin(c[k], ba.c + w); ba.w += w; }
    Agg fi = fa, bi = ba;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Agg pf = { __shfl_up(fi.c, o), __shfl_up(fi.w, o) };
        Agg pb = { __shfl_down(bi.c, o), __shfl_down(bi.w, o) };
        if (lane >= o) fi = combine_after(pf, fi);
        if (lane + o < 64) bi = combine_after(pb, bi);
    }
    if (lane == 63) S.wf[wave] = fi;
    if (lane == 0) S.wb[wave] = bi;
    __syncthreads();
    Agg ef = { __shfl_up(fi.c, 1), __shfl_up(fi.w, 1) };
    if (lane == 0) ef = { FCPP_INF, 0.0 };
    Agg eb = { __shfl_down(bi.c, 1), __shfl_down(bi.w, 1) };
    if (lane == 63) eb = { FCPP_INF, 0.0 };
    Agg pre = { FCPP_INF, 0.0 }, suf = { FCPP_INF, 0.0 };
    for (int q = 0; q < wave; ++q) pre = combine_after(pre, S.wf[q]);
    for (int q = FNWAVE - 1; q > wave; --q) suf = combine_after(suf, S.wb[q]);
    ef = combine_after(pre, ef);
    eb = combine_after(suf, eb);
    const double carry_f = S.back.carry, carry_b = S.fwd.carry;
    double uf = fmin(ef.c, carry_f + ef.w), ub = fmin(eb.c, carry_b + eb.w);
    double vf[FIPT];   // first the swept u = (v/3.6)^2, then the final speed in km/h
#pragma unroll
    for (int k = 0; k < FIPT; ++k) { uf = fmin(c[k], uf + wk(k)); vf[k] = uf; }
#pragma unroll
    for (int k = FIPT - 1; k >= 0; --k) { ub = fmin(c[k], ub + wk(k + 1)); vf[k] = fmin(vf[k], ub); }
    const double b_first = ub;   // backward value at this thread's first item
    bool uniform = straight || turn_run;   // one primitive kind; every item (and the previous point) still at its nominal speed
#pragma unroll
    for (int k = 0; k < FIPT; ++k) {
        if (vf[k] < c[k]) { vf[k] = sqrt(vf[k]) * 3.6; uniform = false; }     // slowed by a sweep
        else if ((clmask >> k) & 1u) {                                         // untouched: exactly the clamped value
            bool cl;
            vf[k] = clamped_speed(nom_v(nom, fs[k]), kap[k], cst, cl);
            uniform = false;
        } else vf[k] = (straight || turn_run) ? vn_run : nom_v(nom, fs[k]);   // untouched: exactly the nominal value
    }

    FCPP_STAMP(6);
    // ---- 5. previous point's final v / kappa / nominal v (for the segment metrics) ------------------
    double vprev = __shfl_up(vf[FIPT - 1], 1), kprev = __shfl_up(kap[FIPT - 1], 1);
    uint32_t fsprev = __shfl_up(fs[FIPT - 1], 1);
    if (lane == 63) { S.ev[wave] = vf[FIPT - 1]; S.ek[wave] = kap[FIPT - 1]; S.efs[wave] = fs[FIPT - 1]; }
    __syncthreads();
    if (lane == 0) {
        if (wave > 0) { vprev = S.ev[wave - 1]; kprev = S.ek[wave - 1]; fsprev = S.efs[wave - 1]; }
        else if (S.back.valid) {
            // final value at s-1: forward part = carry_f, backward part = B(s) + w(s-1,s)
            const double up = fmin(carry_f, b_first + wk(0));
            vprev = (up < S.back.u0) ? sqrt(up) * 3.6 : S.back.v0;
            kprev = S.back.kappa; fsprev = S.back.fs;
        }
    }
    const double vnprev = nom_v(nom, fsprev);

    FCPP_STAMP(7);
    // ---- 6. metrics (MLP:1290-1311) and a_lat validation (MLP:1383-1408) ---------------------------------
    double s_len[2] = { 0, 0 }, s_tpre[2] = { 0, 0 }, s_t[2] = { 0, 0 }, mk = 0, ma = 0, mj = 0;
    int nv = 0;
    uniform = uniform && (vprev == vn_run) && (vnprev == vn_run);
    if (uniform) {
        // one layer, one speed, before and after the speed plan: sum the lengths, divide once
        // (item 0's segment does not count if it starts the path or the headland layer)
        double sd = (at_start || k_seam == 0) ? 0.0 : d[0];
#pragma unroll
        for (int k = 1; k < FIPT; ++k) sd += d[k];
        const int layer = k_seam <= 0 ? 1 : 0;
        const double t = sd / fmax(ms_run, 0.1);
        s_len[layer] = sd; s_tpre[layer] = t; s_t[layer] = t;
    } else {
#pragma unroll
        for (int k = 0; k < FIPT; ++k) {
            if (k < nvalid && !(k == 0 && at_start) && k != k_seam) {      // the seam main|headland belongs to neither layer
                const int layer = k > k_seam ? 1 : 0;
                const double vn_k = nom_v(nom, fs[k]);
                const double vp = k == 0 ? vprev : vf[k - 1];
                const double vnp = k == 0 ? vnprev : nom_v(nom, fs[k - 1]);
                s_len[layer] += d[k];
                // (v + v) / 2 / 3.6 == v / 3.6: equal nominal speeds on both ends take the tabulated m/s value
                const double ms_pre = (vnp == vn_k) ? nom_ms(nom, fs[k]) : ((vnp + vn_k) / 2) / 3.6;
                const double tpre = d[k] / fmax(ms_pre, 0.1);
                s_tpre[layer] += tpre;
                s_t[layer] += (vp == vnp && vf[k] == vn_k) ? tpre : d[k] / fmax(((vp + vf[k]) / 2) / 3.6, 0.1);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < FIPT; ++k) {
        if (k < nvalid && !(k == 0 && at_start) && k != k_end) {     // interior points of the path
            const double kp = k == 0 ? kprev : kap[k - 1];
            if (kap[k] > 0.0) {            // kappa == 0 contributes a_lat = 0: neither a maximum nor a violation
                const double ms = vf[k] / 3.6, alat = ms * ms * kap[k];
                mk = fmax(mk, kap[k]); ma = fmax(ma, alat);
                if (alat > cst.a_lat) { ++nv; fs[k] |= FCPP_FLAG_ALAT; }
            }
            // |kappa_i - kappa_(i-1)| for i >= 2 (MLP:1404-1406)
            if (kap[k] != kp && !(i0 + k == 1)) mj = fmax(mj, fabs(kap[k] - kp));
        }
    }
    put(vo, vf);
    {
        uint32_t *b32 = reinterpret_cast<uint32_t *>(buf);
#pragma unroll
        for (int k = 0; k < FIPT; ++k) b32[2 * lidx(lane * FIPT + k)] = fs[k];
        wave_sync();
#pragma unroll
        for (int m = 0; m < FIPT; ++m) {
            const int p = lane + 64 * m;
            if (p < cw) fso[g0 + p] = b32[2 * lidx(p)];
        }
    }

    FCPP_STAMP(8);
    // ---- 7. fixed-shape block reduction of the metrics ---------------------------------------------
    double dv[9] = { s_len[0], s_tpre[0], s_t[0], s_len[1], s_tpre[1], s_t[1], mk, ma, mj };
    long long iv[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) dv[k] = wave_sum(dv[k]);
#pragma unroll
    for (int k = 6; k < 9; ++k) dv[k] = wave_max0(dv[k]);
    iv[0] = wave_sum_i(nv); iv[1] = wave_sum_i(nout); iv[2] = wave_sum_i(nobs); iv[3] = wave_sum_i(adj);
    if (lane == 0) {
        for (int k = 0; k < 9; ++k) S.R.d[wave][k] = dv[k];
        for (int k = 0; k < 4; ++k) S.R.i[wave][k] = iv[k];
    }
    __syncthreads();
    if (tid == 0) {
        double a[9]; long long b[4];
        for (int k = 0; k < 9; ++k) a[k] = S.R.d[0][k];
        for (int k = 0; k < 4; ++k) b[k] = S.R.i[0][k];
        for (int wv = 1; wv < FNWAVE; ++wv) {
            for (int k = 0; k < 6; ++k) a[k] += S.R.d[wv][k];
            for (int k = 6; k < 9; ++k) a[k] = fmax(a[k], S.R.d[wv][k]);
            for (int k = 0; k < 4; ++k) b[k] += S.R.i[wv][k];
        }
        TilePartial tp;
        tp.main_len = a[0]; tp.main_time_pre = a[1]; tp.main_time = a[2];
        tp.head_len = a[3]; tp.head_time_pre = a[4]; tp.head_time = a[5];
        tp.max_kappa = a[6]; tp.max_alat = a[7]; tp.max_jump = a[8];
        tp.n_viol = b[0]; tp.n_outside = b[1]; tp.n_in_obstacle = b[2]; tp.n_adjusted = b[3];
#ifndef FCPP_DIAG_STAMPS
        partial[tl.stat_tile] = tp;      // (the tile's statistics entry: the entries of a path lie side by side)
#endif
    }
#ifdef FCPP_DIAG_STAMPS
    FCPP_STAMP(9);
    if (tid == 0) {   // lane 0 of the chosen wave: phase durations in shader cycles
        TilePartial tp;
        tp.main_len = (double)(stamp[1] - stamp[0]); tp.main_time_pre = (double)(stamp[2] - stamp[1]);
        tp.main_time = (double)(stamp[3] - stamp[2]); tp.head_len = (double)(stamp[4] - stamp[3]);
        tp.head_time_pre = (double)(stamp[5] - stamp[4]); tp.head_time = (double)(stamp[6] - stamp[5]);
        tp.max_kappa = (double)(stamp[10] - stamp[0]); tp.max_alat = (double)(stamp[11] - stamp[10]); tp.max_jump = (double)(stamp[1] - stamp[11]);
        tp.n_viol = (long long)(stamp[7] - stamp[6]); tp.n_outside = (long long)(stamp[8] - stamp[7]);
        tp.n_in_obstacle = (long long)(stamp[9] - stamp[8]); tp.n_adjusted = (long long)(stamp[9] - stamp[0]);
        partial[tl.stat_tile] = tp;
    }
#endif
}

// ---- turn templates (once per batch) -----------------------------------------------------------------------
// tmpl_u[k]: U-turn sample k.  Arcs (MLP:815-823): (R cos th_k, R sin th_k).  Clothoid: (Re * Y_k, Re * X_k) of the
// unit clothoid-arc-clothoid shape, i.e. the world offsets of cac_world_point for the start heading +y.
// tmpl_c[k]: corner sample k = (t1, t2).  Arcs (MLP:1049-1060): (R (1 - cos th_k), R sin th_k); clothoid: (Re * Y_k, Re * X_k).
__global__ void k_build_templates(TurnTemplates tt, const CacShape *__restrict__ shapes, double2 *__restrict__ tu,
                                  double2 *__restrict__ tc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < tt.nu) {
        const double sv = linspace_at(0.0, tt.u_end, tt.u_step, tt.nu, k);
        double2 o;
        if (tt.turn_model == FCPP_TURN_ARC) {
            double sn, cs;
            sincos(sv, &sn, &cs);
            o.x = tt.R * cs; o.y = tt.R * sn;
        } else {
            double X, Y;
            cac_unit_point(shapes[0], sv / tt.u_Re, X, Y);
            o.x = tt.u_Re * Y; o.y = tt.u_Re * X;
        }
        tu[k] = o;
    }
    if (k < tt.nc) {
        const double sv = linspace_at(0.0, tt.c_end, tt.c_step, tt.nc, k);
        double2 o;
        if (tt.turn_model == FCPP_TURN_ARC) {
            double sn, cs;
            sincos(sv, &sn, &cs);
            o.x = tt.R * (1 - cs); o.y = tt.R * sn;
        } else {
            double X, Y;
            cac_unit_point(shapes[1], sv / tt.c_Re, X, Y);
            o.x = tt.c_Re * Y; o.y = tt.c_Re * X;
        }
        tc[k] = o;
    }
}

// the U-turn shape's own metrics, once per batch: dk[k] = (segment length |t_k - t_(k-1)|, curvature of (t_(k-1), t_k, t_(k+1))).
// Every U-turn of the batch is a translate / mirror (/ rotation, for tilted fields) of the template, under which both are
// invariant; taken from the shape itself they carry none of the rounding noise of 1000 m coordinates.
__global__ void k_build_template_metrics(int n, const double2 *__restrict__ t, double2 *__restrict__ dk)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double2 o = make_double2(0.0, 0.0);
    if (k >= 1) {
        const double dx1 = t[k].x - t[k - 1].x, dy1 = t[k].y - t[k - 1].y;
        o.x = seg_len(dx1, dy1);
        if (k + 1 < n) {
            const double dx2 = t[k + 1].x - t[k].x, dy2 = t[k + 1].y - t[k].y;
            o.y = curv_chords(dx1, dy1, o.x, dx2, dy2, seg_len(dx2, dy2));
        }
    }
    dk[k] = o;
}

int launch_field_junctions(hipStream_t st, int64_t n_fields, const DevField *fields, const DevConst &cst, void *junc)
{
    if (n_fields <= 0) return 0;
    hipLaunchKernelGGL(k_field_junctions, dim3((unsigned)((n_fields + 255) / 256)), dim3(256), 0, st, n_fields, fields, cst, (double2 *)junc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_build_template_metrics(hipStream_t st, int n, const void *tmpl, void *dk)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_build_template_metrics, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const double2 *)tmpl, (double2 *)dk);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_build_templates(hipStream_t st, const TurnTemplates &tt, const CacShape *shapes, void *tu, void *tc)
{
    const int n = tt.nu > tt.nc ? tt.nu : tt.nc;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_build_templates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tt, shapes, (double2 *)tu,
                       (double2 *)tc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// kinds: which chunk kinds the list holds -- 14: straights and U-turns, 16: layer-1 spans
int launch_plan_quiet(hipStream_t st, int64_t n_chunks, const DevTile *chunks, int kinds, const DevField *fields, const DevPrim *prims,
                      const DevConst &cst, const DevObstacles &obs, double *x, double *y, double *kappa, double *v, uint32_t *fs,
                      TilePartial *partial)
{
    if (n_chunks <= 0) return 0;
    const bool staged = cst.tmpl_n > 0 && cst.tmpl_n <= TMPL_LDS, has_obs = obs.offsets != nullptr;
    // Chunks (= wavefronts) per workgroup: eight for long launches of the dense instances -- on ten sets of output arrays each
    // (tools/ab_knob.py FCPP_QUIET_WPB=4,8 and a sweep over allocations) cfg2 at 0.5 m 1.202 -> 1.190 ms on its fastest set and ~1 % on all
    // others, the dense kernel of cfg2 at 0.1 m 4772 -> 4705 us, cfg3 394 -> 387 us -- four for launches of a few rounds (the headline's
    // 14 120 chunks: 31.9 us, 34.3 with eight) and for the spans at the reference's sampling (STAGED): cfg5 gains 2-5 % with eight on the
    // slow allocations (2.40 -> 2.33, 2.61 -> 2.55 ms) and loses 1 % on the fast ones (2.045 -> 2.07 ms), which calibration finds; two per
    // workgroup: +11 % on cfg5.
    const int wpb_k = tune_int("FCPP_QUIET_WPB", 0);
    const int wpb = wpb_k == 8 || wpb_k == 4 ? wpb_k : (n_chunks >= 65536 && !(kinds == 16 && staged) ? 8 : 4);
    const dim3 grid((unsigned)((n_chunks + wpb - 1) / wpb)), block(64 * wpb);
    // Resident waves of the span kernel are held to FOUR per SIMD by its LDS footprint (34 KiB per four-wave workgroup of 160 KiB per
    // CU): measured on identical memory (tools/ab_knob.py) cfg5 1.33 vs 1.44 ms at 5-7 waves and 1.59 ms at 3; the other
    // configurations do not care (+-1 %) -- except launches of a few rounds of workgroups, where FIVE per SIMD (27 KiB) end a round
    // earlier: the headline's 3530 workgroups 31.9 vs 33.2 us, step 0.0819 vs 0.0837 ms; cfg2 at the reference's sampling, 2064
    // workgroups, the same either way -- until round 3 put this kernel first in the step: since then four per SIMD are the faster choice
    // there too (headline 31.2 vs 31.8 us, step 0.0679 vs 0.0685 ms; cfg2 at the reference's sampling 23.4 vs 23.6 us).
    // FCPP_SPAN_LDS / FCPP_QUIET_PAD: bytes, for that tool.
    const int static_lds = (has_obs ? wpb * 2 * OBS_LDS_VERTS * 8 : 32) + (kinds == 16 && staged ? wpb * 3 * TMPL_LDS * 8 : 32);
    const int span_lds = 34 * 1024 * wpb / 4;
    const int pad = std::min(64 * 1024, kinds == 16 ? std::max(0, tune_int("FCPP_SPAN_LDS", span_lds) - static_lds) : std::max(0, tune_int("FCPP_QUIET_PAD", 0)));
#define FCPP_QUIET(K, SD, TL, OB) do { if (wpb == 8) FCPP_LAUNCH((k_plan_quiet<K, SD, TL, OB, 8>), grid, block, pad, st, chunks, fields, prims, cst, obs, x, y, kappa, v, fs, partial, n_chunks); \
                                       else FCPP_LAUNCH((k_plan_quiet<K, SD, TL, OB, 4>), grid, block, pad, st, chunks, fields, prims, cst, obs, x, y, kappa, v, fs, partial, n_chunks); } while (0)
    // Instances: spans fetch their descriptors by scalar loads (SCALAR_DESC: that nearly halved their time in round 1), stage the
    // turn template in LDS when it is the reference's short one (STAGED), and every instance exists with and without the polygon
    // tests (OBS; without: 68 instead of 93 vector registers and no scalar spills in the span kernel, 80 instead of 116 in the dense
    // one -- cfg2 at 0.5 m 1.63 -> 1.39 ms).
    if (kinds == 16 && staged && !has_obs) FCPP_QUIET(16, true, true, false);
    else if (kinds == 16 && staged) FCPP_QUIET(16, true, true, true);
    else if (kinds == 16 && !has_obs) FCPP_QUIET(16, true, false, false);
    else if (kinds == 16) FCPP_QUIET(16, true, false, true);
    else if (!has_obs) FCPP_QUIET(14, false, false, false);
    else FCPP_QUIET(14, false, false, true);
#undef FCPP_QUIET
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_plan_fused(hipStream_t st, int variant, int64_t n_tiles, const int32_t *ids, const DevTile *tiles,
                      const DevField *fields, const DevPrim *prims, const DevConst &cst, const DevObstacles &obs, double *x,
                      double *y, double *kappa, double *v, uint32_t *fs, TilePartial *partial)
{
    if (n_tiles <= 0) return 0;
    // variant = register budget: minimum waves per SIMD the compiler must allow (3 -> <=168 VGPRs, 4 -> <=128, 2 -> <=256)
    if (variant == 4)
        FCPP_LAUNCH(k_plan_fused<4>, dim3((unsigned)n_tiles), dim3(FBLOCK), 0, st, tiles, fields, prims, cst, obs, x, y,
                           kappa, v, fs, partial, ids);
    else if (variant == 2)
        FCPP_LAUNCH(k_plan_fused<2>, dim3((unsigned)n_tiles), dim3(FBLOCK), 0, st, tiles, fields, prims, cst, obs, x, y,
                           kappa, v, fs, partial, ids);
    else
        FCPP_LAUNCH(k_plan_fused<3>, dim3((unsigned)n_tiles), dim3(FBLOCK), 0, st, tiles, fields, prims, cst, obs, x, y,
                           kappa, v, fs, partial, ids);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fcpp
