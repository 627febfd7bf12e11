#!/usr/bin/env python3
"""Measurement of the GA evolution loop on the device (fcpp_ga_evolve): D from 128 points U[0,1000)^2
(seed 128), population 4096, 500 generations (the convergence test is disabled so that all of them run).  One JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402
from field_coverage_path_planning_amd.genetic_algorithm_solver import GAConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, default=128)
ap.add_argument('--pop', type=int, default=4096)
ap.add_argument('--generations', type=int, default=500)
ap.add_argument('--no-cpu-baseline', action='store_true')
ap.add_argument('--reps', type=int, default=7)
a = ap.parse_args()
rng = np.random.default_rng(128)
pts = rng.uniform(0, 1000, size=(a.nodes, 2))
D = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
routes = np.array([rng.permutation(a.nodes) for _ in range(a.pop)], dtype=np.int32)
cfg = GAConfig(population_size=a.pop, max_generations=a.generations, convergence_threshold=10 ** 9)
Dd = torch.as_tensor(D, device='cuda')
rd = torch.as_tensor(routes, device='cuda')
E.ga_evolve(Dd, rd, GAConfig(population_size=a.pop, max_generations=3), seed=1)      # warm-up
torch.cuda.synchronize()
dts = []
for _ in range(a.reps):                 # (one run is ~7 ms: too short to judge alone; the median of a few is reported, all are listed)
    r0 = rd.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    final, best, hb, ha, res = E.ga_evolve(Dd, r0, cfg, seed=4096)
    torch.cuda.synchronize()
    dts.append(time.perf_counter() - t0)
dt = sorted(dts)[len(dts) // 2]
evals = (res.generations + 1) * a.pop
out = {'metric': 'GA chromosome evaluations/s, whole generations on the device (fcpp_ga_evolve)', 'unit': 'chromosomes/s',
       'value': evals / dt, 'config': {'workload': f'cfg4: n={a.nodes}, population {a.pop}, {res.generations} generations', 'seed': 4096},
       'seconds': dt, 'us_per_generation': dt / max(res.generations, 1) * 1e6,
       'us_per_generation_each_rep': [round(d / max(res.generations, 1) * 1e6, 2) for d in dts], 'best_distance': res.best_distance,
       'initial_best_distance': float(1 / hb[0] - 1e-6) if len(hb) else None}
if not a.no_cpu_baseline:
    import oracle as orc
    g = max(1, min(a.generations, 12))
    t0 = time.perf_counter()
    orc.ga_evolve(D, routes, population_size=a.pop, max_generations=g, convergence_threshold=10 ** 9, seed=4096)
    dc = time.perf_counter() - t0
    out['cpu_baseline'] = {'value': (g + 1) * a.pop / dc, 'unit': 'chromosomes/s', 'cores': 1, 'kind': 'port',
                           'sample': f'{g} generations through oracle/fcpp_oracle.c orc_ga_evolve ({dc:.2f} s)'}
print(json.dumps(out))
