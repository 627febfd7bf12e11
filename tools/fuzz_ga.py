#!/usr/bin/env python3
"""Randomised bit-exactness sweep of fcpp_ga_evolve against the oracle's replay (orc_ga_evolve): random problem sizes, populations,
elite counts, tournament sizes, rates and seeds -- all three device paths (one launch per generation, two streams, the general elite
selection).   fuzz_ga.py [--seconds 120] [--seed 1]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

import oracle as orc  # noqa: E402
from field_coverage_path_planning_amd import engine as E  # noqa: E402
from field_coverage_path_planning_amd.genetic_algorithm_solver import GAConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=120.0)
ap.add_argument('--seed', type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, runs, fails = time.time(), 0, 0
while time.time() - t0 < a.seconds:
    n = int(rng.choice([2, 3, 7, 33, 64, 65, 128, 129, 200, 400, 520, 700]))
    pop = 2 * int(rng.integers(2, 40)) if rng.random() < 0.7 else 2 * int(rng.integers(40, 3300))
    if n > 300:
        pop = min(pop, 128)
    elite = int(rng.integers(0, min(pop - 1, 80)))
    tour = int(rng.integers(1, min(pop, 64) + 1))
    cfg = dict(max_generations=int(rng.integers(1, 40 if pop * n < 200000 else 6)), elite_size=elite, tournament_size=tour,
               convergence_threshold=int(rng.choice([3, 8, 1000])), crossover_rate=float(rng.choice([0.0, 0.5, 0.85, 1.0])),
               mutation_rate=float(rng.choice([0.0, 0.02, 0.5, 1.0])))
    seed = int(rng.integers(0, 2 ** 62))
    pts = rng.uniform(0, 1000, size=(n, 2))
    D = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
    if rng.random() < 0.3:
        D = np.round(D)                       # ties in fitness
    routes = np.array([rng.permutation(n) for _ in range(pop)], dtype=np.int32)
    if rng.random() < 0.3:
        routes[pop // 2:] = routes[:pop - pop // 2]          # duplicate chromosomes: equal fitness values, index tie-breaks
    c = GAConfig(population_size=pop, **cfg)
    final, best, hb, ha, res = E.ga_evolve(D, routes, c, seed=seed)
    ofinal, obest, ohb, oha, ores = orc.ga_evolve(D, routes, population_size=pop, seed=seed, **cfg)
    ok = (res.generations, res.convergence_gen) == (ores.generations, ores.convergence_gen) and res.best_distance == ores.best_distance and \
        np.array_equal(final.cpu().numpy(), ofinal) and np.array_equal(best.cpu().numpy(), obest) and np.array_equal(hb, ohb) and np.array_equal(ha, oha)
    if not ok:
        fails += 1
        print('MISMATCH', dict(n=n, pop=pop, seed=seed, **cfg), flush=True)
    runs += 1
    if runs % 50 == 0:
        print(f'{runs} runs, {fails} mismatches, {time.time() - t0:.0f} s', flush=True)
print(f'DONE {runs} runs, {fails} mismatches')
sys.exit(1 if fails else 0)
