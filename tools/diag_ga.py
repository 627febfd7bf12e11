#!/usr/bin/env python3
"""Diagnostic only: where the time of one GA generation goes (cfg4: 128 nodes, population 4096).  Needs the -DFCPP_DIAG_GA build
(`make diag-ga` -> build/libfcpp_diag_ga.so, selected with FCPP_LIBRARY); prints the phase stamps (10 ns ticks) of pair 1000's wavefront in
generation 250 and of the two bookkeeping workgroups of the launch before it, and the loop's wall clock per generation."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from field_coverage_path_planning_amd import engine as E, _lib as L, workloads as WL

D, routes = WL.cfg4_ga()


class Cfg:
    population_size, max_generations, crossover_rate, mutation_rate = 4096, 500, 0.85, 0.02
    elite_size, tournament_size, convergence_threshold = 20, 5, 10 ** 9


Dd = torch.as_tensor(D, device='cuda')
E.ga_evolve(Dd, routes, Cfg, seed=4096)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    E.ga_evolve(Dd, routes, Cfg, seed=4096)
torch.cuda.synchronize()
print('us per generation (wall clock, 3 runs of 500): %.2f' % ((time.perf_counter() - t0) / 3 / 500 * 1e6))
out = (C.c_uint64 * 48)()
L.load().fcpp_diag_ga_stamps(out)
st = list(out)
names = [(8, 'workgroup entry (flag asked for)'), (0, 'pair: entry'), (1, 'pair: two tournaments decided (Philox, candidates\' fitness, arg-max)'), (2, 'pair: parents\' genes in LDS'),
         (3, 'pair: order crossover of both children'), (4, 'pair: mutation'), (5, 'pair: matrix entries asked for, rows stored'),
         (6, 'pair: terms in LDS'), (7, 'pair: left-to-right sums, fitness written')]
prev = st[8] or st[0]
t00 = prev
for k, n in names:
    if st[k]:
        print(f'{n:76s} +{(st[k] - prev) * 10:6d} ns   at {(st[k] - t00) * 10:6d} ns')
        prev = st[k]
print('bookkeeping workgroup (statistics, best-so-far): %d ns; elites workgroup: %d ns' % ((st[17] - st[16]) * 10, (st[19] - st[18]) * 10))
