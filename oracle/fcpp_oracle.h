/*
 * fcpp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's hot path
 * (qwagrox/field-coverage-path-planning, multi_layer_planner_v3.py = "MLP",
 * genetic_algorithm_solver.py = "GA").  It is the CHECKER for the HIP library:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  Nothing under field_coverage_path_planning_amd/ links, imports or calls it.
 *
 * Pinning: tests/test_oracle_vs_golden.py checks every function below against
 * the .npz files in tests/golden, which tools/gen_golden.py produced by running the
 * reference's own Python code in the build container.
 *
 * NOT pinned by the reference ("parity unpinned"): the clothoid turn
 * model, dense resampling, the point-in-polygon geofence / obstacle flags and
 * every Shapely/GEOS-dependent value.  Those parts are marked
 * BUILD-DEFINED below.
 */
#ifndef FCPP_ORACLE_H
#define FCPP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* VehicleParams, MLP:29-39 (same field order) */
typedef struct {
    double working_width, min_turn_radius, max_work_speed_kmh, max_headland_speed_kmh,
        headland_turn_speed_kmh, max_lateral_accel, max_longitudinal_accel, safety_factor;
} orc_vehicle;

/* BUILD-DEFINED sampling options; all-zero = the reference's behaviour */
typedef struct {
    int32_t turn_model;     /* 0 arcs (MLP:807-825,1046-1062), 1 clothoid-arc-clothoid */
    int32_t clothoid_fit;   /* 0 keep kappa_max = 1/R ; 1 keep the reference arc's end point (chord) */
    double sample_spacing;  /* 0 = reference counts 2/20/15/20 ; >0 = uniform arc-length spacing [m] */
    double clothoid_frac;   /* share of a turn's heading change spent in the two clothoids, [0,1] */
    double geofence_tol;    /* a point is out of field if it is more than this outside [m] */
    int32_t obstacle_mode;  /* 0 obstacles only flag points (the reference, MLP:731-732) ; 1 swaths clipped and re-routed (include/fcpp.h) */
    int32_t ring_order;     /* order in which `buffer(-d).exterior.coords[:-1]` lists the inset corners (MLP:964-972), a GEOS fact the reference
                               never pins: 0 = as the field vertices (the documented intent LL, LR, UR, UL of MLP:957, 1049-1058);
                               1 = the opposite direction from the same first vertex (LL, UL, UR, LR: a clockwise shell) */
} orc_options;

typedef struct {
    double vx[4], vy[4];         /* field vertices; rectangle = (0,0),(L,0),(L,H),(0,H)  MLP:127-132 */
    int32_t from_vertices;       /* 1: constructed with field_vertices=, 0: field_length/field_width */
    int32_t has_start, has_end;  /* as passed by the caller (before _validate_point) */
    double start_x, start_y, end_x, end_y;
    int32_t n_obstacles;         /* obstacle polygons (validator only; MLP:731 ignores them) */
    const int64_t *obs_offsets;  /* n_obstacles+1 */
    const double *obs_xy;        /* AoS vertices */
} orc_field;

/* flag/segment word, one per path point (same packing as include/fcpp.h) */
enum {
    ORC_KIND_SWATH = 0, ORC_KIND_UTURN = 1, ORC_KIND_HEAD_START = 2, ORC_KIND_HEAD_STRAIGHT = 3,
    ORC_KIND_CORNER = 4, ORC_KIND_REVERSE = 5, ORC_KIND_DETOUR = 6,
    ORC_KIND_MASK = 7u, ORC_FLAG_HEADLAND = 8u, ORC_FLAG_ALAT = 16u, ORC_FLAG_OUTSIDE = 32u,
    ORC_FLAG_OBSTACLE = 64u, ORC_INDEX_SHIFT = 8
};

typedef struct {
    /* sizes / decisions (integers: bit-exact parity) */
    int64_t n_main, n_head;
    int32_t n_swaths, n_loops, start_corner, reverse_order, start_from_right, rotated;
    int32_t start_kept, end_kept, shape; /* shape: 0 rectangle 1 parallelogram 2 other */
    int32_t n_reverse[4];                /* reverse points appended at corner index c (outer loop), 0 if none */
    double corner_angles[4], field_length, field_width, headland_width, rotation_angle;
    /* concatenated path main||headland, AoS like the reference (N x 2) */
    double *xy, *v, *kappa;
    uint32_t *flagseg;
    /* stats, MLP:616-628, 882-895, 423-431 */
    double main_len_m, main_time_pre_s, main_time_s, head_len_m, head_time_pre_s, head_time_s;
    /* verify_curvature_constraints over the concatenation, MLP:1373-1424 */
    double max_kappa, max_alat, viol_rate, max_jump;
    int64_t n_viol, n_outside, n_in_obstacle, n_adjusted;
    int32_t pass;
    /* connectors, MLP:1313-1355 (50 x 2 each) */
    int32_t has_approach, has_departure;
    double approach[100], departure[100];
} orc_plan;

/* ---- reference restatements ------------------------------------------------ */
double orc_curvature(const double *p1, const double *p2, const double *p3);              /* MLP:513-536 */
void orc_smooth_speed_profile(const double *xy, double *v, int64_t n, double a_lon);     /* MLP:538-589 */
int64_t orc_speed_limit(const double *xy, const double *v_in, double *v_out, int64_t n,
                        const orc_vehicle *veh);                                         /* MLP:467-511 */
void orc_verify(const double *xy, const double *v, int64_t n, const orc_vehicle *veh,
                double *out6 /* max_kappa,max_alat,n_viol,rate,max_jump,pass */);        /* MLP:1373-1424 */
double orc_path_length(const double *xy, int64_t n);                                     /* MLP:1290-1296 */
double orc_work_time(const double *xy, const double *v, int64_t n);                      /* MLP:1298-1311 */
void orc_linspace(double a, double b, int64_t n, double *out);                           /* numpy.linspace */
void orc_safe_arc_turn(double y, int turn_right, double min_x, double max_x, double R,
                       double *xy20);                                                    /* MLP:791-830 */
void orc_corner_arc(double cx, double cy, int corner_index, double R, int n, double *xy); /* MLP:1580-1608 */
void orc_straight(double x0, double y0, double x1, double y1, int64_t n, double *xy);    /* MLP:1013-1022 */
void orc_rotate_point(double x, double y, double ang, double cx, double cy, double *out);/* MLP:265-284 */
double orc_distance_to_boundary(double x, double y, double dx, double dy, double L, double H,
                                double R);                                               /* MLP:1220-1288 */
int64_t orc_reverse_path(const double *end, const double *second_last, double L, double H, double R,
                         double spacing, double *len_out, double *xy /* may be NULL */);  /* MLP:1154-1218 */
int64_t orc_u_pattern(double min_x, double min_y, double max_x, double max_y, int reverse_order,
                      int start_from_right, const orc_vehicle *veh, double *xy, double *v,
                      int64_t cap);                                                      /* MLP:720-789 */
double orc_ga_distance(const int32_t *route, int32_t n, const double *D);                /* GA:174-181 */
double orc_ga_fitness(const int32_t *route, int32_t n, const double *D);                 /* GA:168-172 */

/* whole plan, MLP:63-107 + 387-465; returns 0 or a negative error (-1 bad input = the
 * reference's ValueError, -2 headland loop empty, -3 unsupported polygon) */
int orc_plan_field(const orc_field *f, const orc_vehicle *veh, const orc_options *opt, orc_plan *out);
void orc_plan_free(orc_plan *p);

/* ---- BUILD-DEFINED pieces ---------------------------------------------------- */
void orc_fresnel(double t, double *C, double *S);  /* C(t)=int_0^t cos(pi u^2/2) du, long-double quadrature */
/* clothoid-arc-clothoid turn: start pose (x0,y0,th0), signed heading change dth, effective radius Re,
 * clothoid share f; returns total length; point at arc length s via orc_cac_point */
double orc_cac_length(double dth, double Re, double f);
void orc_cac_point(double x0, double y0, double th0, double dth, double Re, double f, double s,
                   double *out_xy);
double orc_cac_fit_radius(double dth, double R, double f, int fit);
int orc_point_in_polygon(double px, double py, const double *poly_xy, int64_t nv);  /* even-odd */
int orc_corner_gap_decision(double R, double W);   /* MLP:1070 `gap.area > 0.1`: 1 / 0 / -1 (undecidable without GEOS) */
/* MLP:599-609, 288: centroid of main_boundary.difference(union of the obstacles' W/2 buffers) for convex obstacles inside the boundary
 * with disjoint grown boxes (GEOS' polygonal buffer restated); 0 = the case is not covered, *cx / *cy untouched */
int orc_difference_centroid(const double *mx, const double *my, const orc_field *f, double r, double *cx, double *cy);
int orc_outside_polygon(double px, double py, const double *poly_xy, int64_t nv, double tol);  /* fcpp_validate's geofence rule (build-defined) */
int orc_outside_convex(double px, double py, const double *vx, const double *vy, int nv, double tol);

/* ---- GA evolution operators (GA:183-268) ------------------------------------------------------------------------------
 * The reference draws from the unseeded stdlib `random`; here every random decision is an explicit input, so the
 * operators are pure functions that the reference's own methods pin (tools/gen_golden.py replays the same decisions
 * through `random.sample` / `random.random`).  orc_ga_evolve draws the decisions from Philox4x32-10 exactly as the HIP
 * kernels do (include/fcpp.h, fcpp_ga_evolve), so whole runs are comparable bit for bit. */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);   /* Salmon et al., SC'11: 10 rounds */
/* GA:183-196: cand = pop x k tournament candidates (distinct per row); selected row s = population[first argmax of fitness] */
void orc_ga_selection(const int32_t *population, const double *fitness, int32_t pop, int32_t n, const int32_t *cand, int32_t k,
                      int32_t *selected);
/* GA:212-242: children of one pair for the cut points a < b */
void orc_ga_ox(const int32_t *p1, const int32_t *p2, int32_t n, int32_t a, int32_t b, int32_t *c1, int32_t *c2);
/* GA:254-268: new_population[pop-e+j] = old_population[argsort(old_fitness)[pop-e+j]] (ties: the larger index is later) */
void orc_ga_elitism(const int32_t *old_population, const double *old_fitness, int32_t pop, int32_t n, int32_t e,
                    int32_t *new_population);
typedef struct orc_ga_config {
    int32_t population_size, max_generations;
    double crossover_rate, mutation_rate;
    int32_t elite_size, tournament_size, convergence_threshold, _pad;
    uint64_t seed;
} orc_ga_config;
typedef struct orc_ga_result { int32_t generations, convergence_gen; double best_distance, best_fitness; } orc_ga_result;
/* GA:64-115 with Philox decisions; routes: in = initial population, out = final; hist (may be NULL): 2 * max_generations */
void orc_ga_evolve(int32_t n, const orc_ga_config *cfg, const double *D, int32_t *routes, int32_t *best_route, double *hist,
                   orc_ga_result *res);

/* MVP:229-259 / MFP:263-288 */
void orc_distance_matrix(int32_t n, const double *x, const double *y, double *D);
/* MFP:290-320: first shortest (exit, entry) pair; returns the distance, indices in *bf / *bt (-1 if a list is empty) */
double orc_best_connection(const double *fx, const double *fy, int64_t nf, const double *tx, const double *ty, int64_t nt,
                           int64_t *bf, int64_t *bt);

/* coverage rasterisation: the sampled restatement of MLP:1426-1509 (corner grids) and MLP:1357-1371 (coverage rate).
 * Sample (i, j) = (ox + (i + shift) * res, oy + (j + shift) * res); covered by a polyline iff within `radius` of one of its
 * segments (strict: <, else <=; division-free test, see include/fcpp.h); polyline B is tried only on samples A left open
 * (MLP:1489-1497).  region: NULL, or 24 doubles = 4 outer + 4 inner half-planes (a, b, c), a sample counts iff inside all
 * outer ones and not inside all inner ones.  grid (may be NULL): nx*ny bytes [j][i], bit 0 = A, bit 1 = B; counts[3] =
 * samples in region, covered by A, covered by A or B. */
void orc_cover_grid(double ox, double oy, double res, double shift, double radius, int32_t nx, int32_t ny, const double *ax,
                    const double *ay, int32_t n_a, const double *bx, const double *by, int32_t n_b, int strict,
                    const double *region, uint8_t *grid, int64_t *counts);

#ifdef __cplusplus
}
#endif
#endif
