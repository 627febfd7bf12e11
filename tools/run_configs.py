#!/usr/bin/env python3
"""Run the BASELINE.json configs other than the bench workload once and print their throughput (1 GPU).
cfg3: 5000x2000 m, 32 convex obstacles, 0.05 m spacing;  cfg4: GA fitness 128 nodes x pop 4096 x 501 evaluations;
cfg5: 65536 parallelograms (reference sampling);  cfg1x: 4096 copies-with-jitter of the 500x200 field at reference sampling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from field_coverage_path_planning_amd import engine as E  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cfg3():
    rng = np.random.default_rng(32)
    obstacles = []
    for gy in range(4):
        for gx in range(8):
            cx = (gx + 0.5) * 5000 / 8 + rng.uniform(-100, 100)
            cy = (gy + 0.5) * 2000 / 4 + rng.uniform(-100, 100)
            r = rng.uniform(10, 40)
            obstacles.append([(cx + r * np.cos(a), cy + r * np.sin(a)) for a in np.arange(8) * np.pi / 4])
    spec = E.FieldSpec(field_length=5000.0, field_width=2000.0, obstacles=obstacles)
    for tm in (0, 1):
        b = E.Batch([spec], E.make_vehicle(), E.make_options(tm, 0.05))
        bufs = b.alloc()
        dt = timed(lambda: b.run(bufs))
        st = b.run(bufs).stats()
        q, g = b.point_split()
        print(f'cfg3 turn_model={tm}: {b.total_points} points, {dt*1e3:.2f} ms, {b.total_points/dt:.3e} pts/s, quiet share '
              f'{q/b.total_points:.2f}, in_obstacle={int(st["n_in_obstacle"][0])}, outside={int(st["n_outside"][0])}, '
              f'viol={int(st["n_viol"][0])}')
        b.close()


def cfg5(n=65536):
    rng = np.random.default_rng(65536)
    specs = []
    for _ in range(n):
        base, height = rng.uniform(100, 1000, 2)
        ang, rot = np.radians(rng.uniform(60, 120)), rng.uniform(-np.pi / 4, np.pi / 4)
        sx = height / np.tan(ang)
        v = np.array([[0, 0], [base, 0], [base + sx, height], [sx, height]]) @ np.array([[np.cos(rot), np.sin(rot)], [-np.sin(rot), np.cos(rot)]])
        specs.append(E.FieldSpec(field_vertices=[(float(a), float(c)) for a, c in v]))
    t0 = time.perf_counter()
    b = E.Batch(specs, E.make_vehicle(), E.make_options())
    t_create = time.perf_counter() - t0
    bufs = b.alloc()
    dt = timed(lambda: b.run(bufs))
    st = b.run(bufs).stats()
    bad = sum(1 for i in b.info if i.status != 0)
    b.set_profiling(True)
    for _ in range(5):
        b.run(bufs)
    kt, _ = b.stage_times()
    q, g = b.point_split()
    print(f'cfg5 ({n} parallelograms, reference sampling): {b.total_points} points, batch_create {t_create:.2f} s, '
          f'{dt*1e3:.2f} ms/run, {b.total_points/dt:.3e} pts/s, unsupported fields {bad}, viol={int(st["n_viol"].sum())}; '
          f'kernels {{{", ".join(f"{k}: {v:.3f}" for k, v in kt.items())}}} ms, quiet points {q}, general {g}')
    b.close()


def cfg1x(n=4096):
    rng = np.random.default_rng(1)
    specs = [E.FieldSpec(field_length=500.0 + rng.uniform(-5, 5), field_width=200.0 + rng.uniform(-5, 5)) for _ in range(n)]
    b = E.Batch(specs, E.make_vehicle(), E.make_options())
    bufs = b.alloc()
    dt = timed(lambda: b.run(bufs), reps=20)
    print(f'cfg1x ({n} fields ~500x200, reference sampling): {b.total_points} points, {dt*1e3:.3f} ms/run, {b.total_points/dt:.3e} pts/s')
    b.close()


def cfg4():
    rng = np.random.default_rng(128)
    pts = rng.uniform(0, 1000, size=(128, 2))
    D = torch.as_tensor(np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1)), device='cuda')
    routes = torch.stack([torch.randperm(128, device='cuda', dtype=torch.int32) for _ in range(4096)])
    for mode in (0, 1):
        def gen():
            for _ in range(501):
                E.ga_fitness(routes, D, order_mode=mode)
        dt = timed(gen, reps=2)
        print(f'cfg4 GA fitness order_mode={mode}: 501 x 4096 chromosomes x 128 nodes in {dt*1e3:.1f} ms '
              f'= {501*4096/dt:.3e} chromosomes/s, {501*4096*128/dt:.3e} gathers/s')


if __name__ == '__main__':
    which = sys.argv[1:] or ['cfg3', 'cfg5', 'cfg1x', 'cfg4']
    for w in which:
        globals()[w]()
