#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box (not part of the test suite: minutes, not seconds): random rectangles / parallelograms,
obstacles, start / end points, vehicles and sampling options through the fused pipeline, every field compared with the CPU oracle
(tests/test_gpu_parity.py's comparison: coordinates, curvature, speeds, flags, statistics).
    fuzz_parity.py [--seconds 180] [--seed 1] [--mode 1]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from tests import test_gpu_parity as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=180.0)
ap.add_argument('--seed', type=int, default=1)
ap.add_argument('--mode', type=int, default=1, help='1 = fused pipeline, 0 = staged pipeline')
ap.add_argument('--avoid', action='store_true', help='obstacle-aware swaths (every batch has obstacles; layouts the mode refuses must be refused by both)')
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, rounds, fields, fails = time.time(), 0, 0, 0
VEHS = [T.DEFAULT_VP, [4.0, 5.0, 9.0, 20.0, 3.0, 2.5, 3.0, 0.8], [2.4, 6.0, 12.0, 18.0, 5.0, 1.2, 0.6, 0.9], [3.2, 8.0, 9.0, 15.0, 4.0, 2.0, 0.05, 0.85]]
while time.time() - t0 < a.seconds:
    seed = int(rng.integers(1, 1 << 30))
    n = int(rng.integers(3, 24))
    para, obst = [False, True, 'quad'][int(rng.integers(0, 3))], a.avoid or bool(rng.integers(0, 3) == 0)
    specs, ofs = T._random_fields(seed, n, para=para, with_obstacles=obst)
    tm = int(rng.integers(0, 2))
    sp = float(rng.choice([0.0, 0.0, 2.0, 1.0, 0.7, 0.5, 0.4, 0.33, 0.27, 0.25, 0.22, 0.2, 0.15, 0.1]))
    opt = dict(turn_model=tm, sample_spacing=sp, ring_order=int(rng.integers(0, 2)))
    if a.avoid:
        opt['avoid_obstacles'] = True
    if tm:
        opt['clothoid_frac'] = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
    veh = VEHS[int(rng.integers(0, len(VEHS)))]
    ds = sp or 0.5
    k_tol = max(T.K_TOL, 4e-12 / ds ** 2 * 4)
    try:
        T._compare_with_oracle_mode(a.mode, specs, ofs, veh, opt, T.XY_TOL, k_tol, max(T.V_TOL, 300 * k_tol))
    except AssertionError as e:
        fails += 1
        print('MISMATCH seed', seed, 'n', n, 'para', para, 'obst', obst, 'opt', opt, 'veh', veh, '::', str(e)[:300], flush=True)
    rounds += 1
    fields += n
    if rounds % 25 == 0:
        print(f'{rounds} batches, {fields} fields, {fails} mismatches, {time.time() - t0:.0f} s', flush=True)
print(f'DONE {rounds} batches, {fields} fields, {fails} mismatches')
sys.exit(1 if fails else 0)
