"""Obstacle-aware swaths (build-defined, include/fcpp.h), CPU only: properties of the oracle's restatement on random obstacle polygons, and the
host planner's sizing against it.  The GPU parity tests (tests/test_gpu_parity.py) compare whole paths; here the checker itself is checked:
every accepted plan keeps its path out of the obstacles, every layer-1 detour point keeps W/2 from every obstacle, and the library's host-side
sizing (fcpp_plan_count: no GPU) gives the oracle's point counts and refusals."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402
from field_coverage_path_planning_amd import _lib as L, engine as E  # noqa: E402


def _dist_to_polygon(px, py, poly):
    q = np.asarray(poly, dtype=np.float64)
    d = np.full(px.shape, np.inf)
    for a, c in zip(q, np.roll(q, -1, axis=0)):
        e = c - a
        t = np.clip(((px - a[0]) * e[0] + (py - a[1]) * e[1]) / max(e @ e, 1e-300), 0.0, 1.0)
        d = np.minimum(d, np.hypot(px - (a[0] + t * e[0]), py - (a[1] + t * e[1])))
    return d


def _random_polygon(rng, cx, cy, r, n):
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(0.35, 1.0, n) * r                       # star-shaped: simple, usually not convex
    return [(float(cx + a * np.cos(t)), float(cy + a * np.sin(t))) for a, t in zip(rad, ang)]


def test_random_obstacle_polygons_clearance_and_host_sizing():
    rng = np.random.default_rng(2024)
    W = 3.2
    accepted = refused = 0
    for trial in range(160):
        obs = []
        for _ in range(int(rng.integers(1, 4))):
            # (off the half-metre grid of the detour legs' sample counts)
            obs.append(_random_polygon(rng, rng.uniform(60, 340) + 0.137, rng.uniform(40, 180) + 0.071, rng.uniform(3, 14), int(rng.integers(3, 9))))
        sp = float(rng.choice([0.0, 0.5, 0.3]))
        tm = int(rng.integers(0, 2))
        rc, p = orc.plan_field(orc.make_field(L=400.0, H=220.0, obstacles=obs), orc.Vehicle.make(), orc.Options.make(tm, 1, sp, 0.5, 1e-6, 1))
        info = E.plan_count([E.FieldSpec(field_length=400.0, field_width=220.0, obstacles=obs)], E.make_vehicle(),
                            E.make_options(tm, sp, avoid_obstacles=True))[0]
        assert info.status == rc, (trial, info.status, rc)
        if rc != 0:
            refused += 1
            continue
        accepted += 1
        assert (info.n_main, info.n_head) == (p.n_main, p.n_head), trial
        assert p.n_in_obstacle == 0, trial
        fs = p.flagseg
        det = ((fs & L.KIND_MASK) == L.KIND_DETOUR) & ((fs & L.FLAG_HEADLAND) == 0)
        if det.any():
            for poly in obs:
                assert _dist_to_polygon(p.xy[det, 0], p.xy[det, 1], poly).min() >= W / 2 - 1e-6, trial
    assert accepted >= 110 and accepted + refused == 160
