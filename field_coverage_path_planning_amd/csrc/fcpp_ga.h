// fcpp_ga.h -- device state and launchers of the GA evolution kernels (fcpp_ga.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/fcpp.h"

namespace fcpp {

struct GaState {
    double best_fit, best_dist;
    int32_t gwi;            // generations without improvement
    int32_t generations;    // generation + 1 of the last completed generation
    int32_t converged;
    int32_t _pad;
    // fitness of the last (elite_size-th) elite of the previous generation: elitism carries the elites over, so the elites of the
    // next population are among its last elite_size rows and the members strictly above this value (0 before the first selection: everybody)
    double elite_thr;
    // round 5b: a word of the host's pinned memory (or null) the bookkeeping role writes after every generation it completes --
    // (converged << 32) | generations, one 8-byte store -- so that fcpp_ga_evolve follows the run without a copy command or a drained
    // stream between the generations' launches (it used to drain the stream every 32 generations: a bubble of ~20 us each time)
    unsigned long long *mirror;
};

constexpr int GA_MAX_NODES = 2048;

int launch_ga_check_perm(hipStream_t st, int n, int pop, const int32_t *routes, int32_t *bad);   // *bad |= 1 unless every row is a permutation
int launch_ga_pairs(hipStream_t st, int n, int pop, const double *D, const int32_t *cur, const double *cur_fit, int32_t *nxt,
                    double *nxt_fit, double *nxt_dist, const fcpp_ga_config &cfg, int gen, const GaState *state);
// stats of the population `cur` (generation gen, -1 = the initial one), then the elites of `cur` into the tail of `nxt`
int launch_ga_stats_elite(hipStream_t st, int n, int pop, const int32_t *cur, const double *cur_fit, const double *cur_dist, int32_t *nxt,
                          double *nxt_fit, double *nxt_dist, const fcpp_ga_config &cfg, int gen, GaState *state, int32_t *best_route,
                          double *hist);
// one generation in one launch: statistics / elites of `cur` (generation index gen - 1) beside its children (generation index gen);
// for tours and populations whose LDS needs fit (ga_generation_fits)
bool ga_generation_fits(int n, int pop);
int launch_ga_generation(hipStream_t st, int n, int pop, const double *D, const int32_t *cur, const double *cur_fit, const double *cur_dist,
                         int32_t *nxt, double *nxt_fit, double *nxt_dist, const fcpp_ga_config &cfg, int gen, GaState *state,
                         int32_t *best_route, double *hist);

}  // namespace fcpp
