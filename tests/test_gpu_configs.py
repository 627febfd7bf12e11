"""BASELINE.json configurations that round 1 left without a GPU test: cfg3 (one 5000 x 2000 m field, 32 obstacles, 0.05 m:
6.3e7 points on ONE path, ~1.2e5 tiles / chunks) and cfg4 (GA: 128 nodes, population 4096), plus the metric's own 4096 x
(500 x 200 m) batch.  Full-size runs are checked by size-independent properties on the device AND against the CPU oracle.
"""
import numpy as np
import pytest
import torch

import oracle as orc
from field_coverage_path_planning_amd import _lib as L
from field_coverage_path_planning_amd import engine as E
from field_coverage_path_planning_amd import workloads as WL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('turn_model', [1, 0])
def test_cfg3_full_size(turn_model):
    (Lf, Hf), obstacles = WL.cfg3_field()
    veh = E.make_vehicle()
    opt = E.make_options(turn_model, 0.05)
    batch = E.Batch([E.FieldSpec(field_length=Lf, field_width=Hf, obstacles=obstacles)], veh, opt)
    n = batch.total_points
    assert n > 6.0e7
    res = batch.run(mode=1)
    dev = res.x.device
    st = res.stats()
    info = batch.info[0]

    # ---- size-independent properties on the device
    fs = res.flagseg.view(torch.int32)
    kind = fs & 7
    nominal = torch.tensor([9.0, 4.0, 15.0, 15.0, 4.0, 2.5, 0.0, 0.0], dtype=torch.float64, device=dev)[kind.long()]
    assert bool((res.v <= nominal).all()) and bool((res.v > 0).all()) and bool((res.kappa >= 0).all())
    n_obs = int(((fs & L.FLAG_OBSTACLE) != 0).sum())
    assert n_obs > 0 and n_obs == int(st['n_in_obstacle'][0])               # the swaths really cross the obstacles
    assert int(((fs & L.FLAG_OUTSIDE) != 0).sum()) == int(st['n_outside'][0])
    assert int(((fs & L.FLAG_ALAT) != 0).sum()) == int(st['n_viol'][0]) == 0
    ms = res.v / 3.6
    alat = ms * ms * res.kappa
    assert float(alat.max()) <= veh.max_lateral_accel * veh.safety_factor ** 2 * (1 + 1e-9)
    np.testing.assert_allclose(float(res.kappa.max()), st['max_kappa'][0], rtol=0, atol=0)
    np.testing.assert_allclose(float(alat.max()), st['max_alat'][0], rtol=1e-12)
    u = ms * ms
    del ms, alat
    dx, dy = res.x[1:] - res.x[:-1], res.y[1:] - res.y[:-1]
    d = torch.sqrt(dx * dx + dy * dy)
    del dx, dy
    slack = (u[1:] - u[:-1]).abs() - 2 * veh.max_longitudinal_accel * d          # the sweeps' defining inequality on every segment
    assert bool(((slack <= 1e-9) | (d < 1e-6)).all())
    del slack, u
    layer = torch.ones(n - 1, dtype=torch.bool, device=dev)
    layer[info.n_main - 1] = False                                              # the seam belongs to neither layer
    np.testing.assert_allclose(st['main_len_m'][0] + st['head_len_m'][0], float((d * layer).sum()), rtol=1e-10)
    del d, layer
    # every obstacle-flagged point really lies inside one of the polygons, and a sample of unflagged ones does not (torch, independent
    # of the kernel's culling / LDS staging): half-plane test of the convex eight-gons
    idx = torch.nonzero((fs & L.FLAG_OBSTACLE) != 0).flatten()
    rng = np.random.default_rng(3)
    probe = torch.cat([idx, torch.as_tensor(rng.integers(0, n, size=2_000_000), device=dev)])
    px, py = res.x[probe], res.y[probe]
    inside = torch.zeros(len(probe), dtype=torch.bool, device=dev)
    for poly in obstacles:
        P = torch.tensor(poly, dtype=torch.float64, device=dev)
        Q = torch.roll(P, -1, 0)
        ex, ey = (Q - P)[:, 0], (Q - P)[:, 1]
        cr = ex[None, :] * (py[:, None] - P[None, :, 1]) - ey[None, :] * (px[:, None] - P[None, :, 0])
        inside |= (cr > 0).all(dim=1) | (cr < 0).all(dim=1)
    flagged = (fs[probe] & L.FLAG_OBSTACLE) != 0
    assert int((inside & ~flagged).sum()) == 0
    # (points exactly on an edge may be flagged by the crossing test and not by the strict half-plane test)
    assert int((flagged & ~inside).sum()) <= 64
    del probe, px, py, inside, flagged, idx

    # ---- a re-run is bit-identical; the staged pipeline agrees
    keep = {k: getattr(res, k).clone() for k in ('x', 'y', 'kappa', 'v')}
    keep_fs, keep_stats = res.flagseg.clone(), res.stats_raw.clone()
    res2 = batch.run(mode=1)
    for k in keep:
        assert torch.equal(getattr(res2, k), keep[k]), k
    assert torch.equal(res2.flagseg, keep_fs) and torch.equal(res2.stats_raw, keep_stats)
    res0 = batch.run(mode=0)
    for k, tol in (('x', 1e-9), ('y', 1e-9), ('kappa', 1e-7), ('v', 1e-7)):
        assert float((getattr(res0, k) - keep[k]).abs().max()) <= tol, k
    assert torch.equal(res0.flagseg, keep_fs)
    st0 = res0.stats()
    np.testing.assert_allclose(st0['main_len_m'], st['main_len_m'], rtol=1e-12)
    np.testing.assert_allclose(st0['main_time_s'], st['main_time_s'], rtol=1e-10)
    assert (st0['n_in_obstacle'][0], st0['n_outside'][0], st0['n_adjusted'][0]) == (st['n_in_obstacle'][0], st['n_outside'][0], st['n_adjusted'][0])
    del res0, res2

    # ---- the whole field against the oracle: every window -- obstacle crossings, tile and chunk edges, turns, the headland
    rc, p = orc.plan_field(orc.make_field(L=Lf, H=Hf, obstacles=obstacles), orc.Vehicle.make(), orc.Options.make(turn_model, 1, 0.05, 0.5))
    assert rc == 0 and p.n == n and (p.n_main, p.n_head) == (info.n_main, info.n_head)
    assert float(np.abs(keep['x'].cpu().numpy() - p.xy[:, 0]).max()) <= 1e-9
    assert float(np.abs(keep['y'].cpu().numpy() - p.xy[:, 1]).max()) <= 1e-9
    k_tol = 4e-12 / 0.05 ** 2 * 4          # the 3-point stencil amplifies coordinate rounding by 4 / ds^2 (ulp(5000) = 9e-13)
    assert float(np.abs(keep['kappa'].cpu().numpy() - p.kappa).max()) <= k_tol
    assert float(np.abs(keep['v'].cpu().numpy() - p.v).max()) <= 200 * k_tol
    got_fs = keep_fs.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_fs, p.flagseg)
    assert (int(st['n_in_obstacle'][0]), int(st['n_outside'][0]), int(st['n_viol'][0]), int(st['n_adjusted'][0])) == \
        (p.n_in_obstacle, p.n_outside, p.n_viol, p.n_adjusted)
    # (6e7 addends: the oracle follows numpy's pairwise summation, the device sums closed-form run lengths in a fixed tree)
    np.testing.assert_allclose([st['main_len_m'][0], st['head_len_m'][0]], [p.main_len_m, p.head_len_m], rtol=1e-10)
    np.testing.assert_allclose([st['main_time_s'][0], st['head_time_s'][0]], [p.main_time_s, p.head_time_s], rtol=1e-9)
    batch.close()


def test_cfg3_with_obstacle_aware_swaths():
    """cfg3 in AVOID mode: the 5000 x 2000 m field, 32 obstacles, 0.05 m -- every swath that meets an obstacle is clipped
    and driven around it; the whole path against the oracle, no point of it inside an obstacle, detours present."""
    (Lf, Hf), obstacles = WL.cfg3_field()
    opt = E.make_options(1, 0.05, avoid_obstacles=True)
    batch = E.Batch([E.FieldSpec(field_length=Lf, field_width=Hf, obstacles=obstacles)], E.make_vehicle(), opt)
    assert batch.info[0].status == 0
    res = batch.run()
    st = res.stats()
    fs = res.flagseg
    assert int(st['n_in_obstacle'][0]) == 0 and int(((fs & L.FLAG_OBSTACLE) != 0).sum()) == 0
    n_detour = int(((fs & L.KIND_MASK) == L.KIND_DETOUR).sum())
    assert n_detour > 10000
    rc, p = orc.plan_field(orc.make_field(L=Lf, H=Hf, obstacles=obstacles), orc.Vehicle.make(), orc.Options.make(1, 1, 0.05, 0.5, 1e-6, 1))
    assert rc == 0 and p.n == batch.total_points and (p.n_main, p.n_head) == (batch.info[0].n_main, batch.info[0].n_head)
    assert float(np.abs(res.x.cpu().numpy() - p.xy[:, 0]).max()) <= 1e-9 and float(np.abs(res.y.cpu().numpy() - p.xy[:, 1]).max()) <= 1e-9
    k_tol = 4e-12 / 0.05 ** 2 * 4
    assert float(np.abs(res.kappa.cpu().numpy() - p.kappa).max()) <= k_tol
    assert float(np.abs(res.v.cpu().numpy() - p.v).max()) <= 200 * k_tol
    assert np.array_equal(fs.cpu().numpy().view(np.uint32), p.flagseg)
    assert (int(st['n_outside'][0]), int(st['n_viol'][0]), int(st['n_adjusted'][0])) == (p.n_outside, p.n_viol, p.n_adjusted)
    np.testing.assert_allclose([st['main_len_m'][0], st['head_len_m'][0]], [p.main_len_m, p.head_len_m], rtol=1e-10)
    batch.close()


def test_cfg1_batch_4096_equals_golden_and_single_field(golden_plans):
    """The metric's own workload: 4096 x (500 x 200 m) in the reference's model.  Every field of the batch equals the reference's
    plan (golden cfg1_500x200) -- position in the batch, tile alignment and chunking must not matter."""
    g = golden_plans
    LH = WL.cfg1_batch(4096)
    batch = E.Batch(WL.specs_from_lh(E, LH), E.make_vehicle())
    res = batch.run()
    n_per = 1256 + 435
    assert batch.total_points == 4096 * n_per
    x, y, v, k = (getattr(res, a).view(4096, n_per) for a in ('x', 'y', 'v', 'kappa'))
    fs = res.flagseg.view(4096, n_per)
    for t in (x, y, v, k, fs):
        assert bool((t == t[0:1]).all())                                   # 4096 identical plans, bit for bit
    st = res.stats()
    for name in ('main_len_m', 'main_time_s', 'head_len_m', 'head_time_s', 'max_kappa', 'max_alat', 'max_jump'):
        assert (st[name] == st[name][0]).all(), name
    n = 'cfg1_500x200'
    xy = np.column_stack([x[4095].cpu().numpy(), y[4095].cpu().numpy()])
    np.testing.assert_allclose(xy[:1256], g[f'{n}/main_path'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(xy[1256:], g[f'{n}/head_path'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v[4095].cpu().numpy(), np.concatenate([g[f'{n}/main_v'], g[f'{n}/head_v']]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(st['main_len_m'][17] / 1000, g[f'{n}/main_stats'][0], rtol=1e-12)
    np.testing.assert_allclose(st['head_len_m'][17] / 1000, g[f'{n}/head_stats'][0], rtol=1e-12)
    assert int(st['n_viol'].sum()) == 0 and int(st['n_outside'].sum()) == 0         # README_en.md:212-215
    res0 = batch.run(mode=0)
    assert torch.equal(res0.flagseg, res.flagseg)
    for a in ('x', 'y', 'v', 'kappa'):
        assert float((getattr(res0, a) - getattr(res, a)).abs().max()) <= 1e-9, a
    batch.close()


def test_cfg4_ga_population_4096_vs_oracle():
    """cfg4 sizes: 128 nodes, population 4096.  A few generations of the device loop equal the oracle's replay bit for bit, and the
    fitness kernel equals the sequential sum on the whole population."""
    D, routes = WL.cfg4_ga()
    d, f = E.ga_fitness(routes, D, order_mode=0)
    assert np.array_equal(d.cpu().numpy(), orc.ga_distance(routes, D))
    assert np.array_equal(f.cpu().numpy(), orc.ga_fitness(routes, D))

    class Cfg:
        population_size, max_generations, crossover_rate, mutation_rate = 4096, 6, 0.85, 0.02
        elite_size, tournament_size, convergence_threshold = 20, 5, 50

    pop, best, hb, ha, res = E.ga_evolve(D, routes, Cfg, seed=4096)
    wpop, wbest, whb, wha, wres = orc.ga_evolve(D, routes, max_generations=6, seed=4096)
    assert np.array_equal(pop.cpu().numpy(), wpop) and np.array_equal(best.cpu().numpy(), wbest)
    assert np.array_equal(hb, whb) and np.array_equal(ha, wha)
    assert (res.generations, res.convergence_gen, res.best_distance, res.best_fitness) == \
        (wres.generations, wres.convergence_gen, wres.best_distance, wres.best_fitness)
    assert all(sorted(r.tolist()) == list(range(128)) for r in pop.cpu().numpy()[::257])

    # the whole configuration: 500 generations (501 population evaluations), never converging early -- and with the reference's default
    # threshold, where the loop stops on its own -- replayed by the oracle bit for bit: final population, best route, both histories
    for threshold in (10 ** 9, 50):
        class Full:
            population_size, max_generations, crossover_rate, mutation_rate = 4096, 500, 0.85, 0.02
            elite_size, tournament_size, convergence_threshold = 20, 5, threshold
        pop, best, hb, ha, res = E.ga_evolve(D, routes, Full, seed=4096)
        wpop, wbest, whb, wha, wres = orc.ga_evolve(D, routes, max_generations=500, convergence_threshold=threshold, seed=4096)
        assert res.generations == wres.generations and (threshold < 10 ** 9 or res.generations == 500)
        assert np.array_equal(pop.cpu().numpy(), wpop) and np.array_equal(best.cpu().numpy(), wbest), threshold
        assert np.array_equal(hb, whb) and np.array_equal(ha, wha), threshold
        assert (res.convergence_gen, res.best_distance, res.best_fitness) == (wres.convergence_gen, wres.best_distance, wres.best_fitness)
        assert res.best_distance < 0.5 * float(orc.ga_distance(routes, D).min())          # it did optimise
    # precondition checks of the entry points
    bad = routes.copy()
    bad[7, 3] = bad[7, 4]
    with pytest.raises(L.FcppError):
        E.ga_evolve(D, bad, Cfg, seed=1)
    bad[7, 3] = 128
    dd, _ = E.ga_fitness(bad, D)
    dd = dd.cpu().numpy()
    assert np.isnan(dd[7]) and np.isfinite(np.delete(dd, 7)).all()


@pytest.mark.parametrize('runs', ['one run per line and turn', 'one span'])
def test_one_path_with_more_than_4096_statistic_entries(runs, monkeypatch):
    """A single field of 2900 passes.  With its lines and U-turns as runs of their own (rounds 2-4; FCPP_DENSE_SPAN=0, and still the rule
    for fields with obstacles) its path holds ~5800 closed-form runs plus the general tiles, more than a workgroup takes in
    k_reduce_stats -- the sliced reduction (64 workgroups + join); as round 5 cuts it, all complete passes are ONE span.  Whole path
    against the oracle either way, statistics included; mixed with small fields so that every class of the reduction runs in one batch."""
    if runs != 'one span':
        monkeypatch.setenv('FCPP_DENSE_SPAN', '0')
    big = dict(L=120.0, H=9300.0)
    small = [dict(L=500.0, H=200.0), dict(L=260.0, H=140.0), dict(L=900.0, H=700.0)]
    for tm, sp in ((0, 0.1), (1, 0.2)):
        fields = [big] + small
        batch = E.Batch([E.FieldSpec(field_length=f['L'], field_width=f['H']) for f in fields], E.make_vehicle(), E.make_options(tm, sp))
        assert batch.info[0].n_swaths > 2800
        res = batch.run(mode=1)
        st = res.stats()
        x, y, kap, v = (getattr(res, k).cpu().numpy() for k in ('x', 'y', 'kappa', 'v'))
        fs = res.flagseg.cpu().numpy().view(np.uint32)
        off = 0
        for k, f in enumerate(fields):
            rc, p = orc.plan_field(orc.make_field(**f), orc.Vehicle.make(), orc.Options.make(tm, 1, sp, 0.5))
            assert rc == 0 and p.n == batch.info[k].n_main + batch.info[k].n_head
            sl = slice(off, off + p.n)
            assert float(np.abs(x[sl] - p.xy[:, 0]).max()) <= 1e-9 and float(np.abs(y[sl] - p.xy[:, 1]).max()) <= 1e-9
            k_tol = 4e-12 / sp ** 2 * 4          # the 3-point stencil amplifies coordinate rounding by 4 / ds^2
            assert float(np.abs(kap[sl] - p.kappa).max()) <= k_tol and float(np.abs(v[sl] - p.v).max()) <= 200 * k_tol
            assert np.array_equal(fs[sl], p.flagseg)
            assert (int(st['n_viol'][k]), int(st['n_outside'][k]), int(st['n_adjusted'][k])) == (p.n_viol, p.n_outside, p.n_adjusted)
            np.testing.assert_allclose([st['main_len_m'][k], st['head_len_m'][k]], [p.main_len_m, p.head_len_m], rtol=1e-11)
            np.testing.assert_allclose([st['main_time_s'][k], st['head_time_s'][k]], [p.main_time_s, p.head_time_s], rtol=1e-10)
            np.testing.assert_allclose([st['max_kappa'][k], st['max_alat'][k], st['max_jump'][k]], [p.max_kappa, p.max_alat, p.max_jump], rtol=1e-9)
            off += p.n
        res0 = batch.run(mode=0)
        st0 = res0.stats()
        np.testing.assert_allclose(st0['main_len_m'], st['main_len_m'], rtol=1e-12)
        np.testing.assert_allclose(st0['head_time_s'], st['head_time_s'], rtol=1e-10)
        assert batch.reduce_classes()[3] == (0 if runs == 'one span' else 1)            # the long path went through the sliced reduction
        batch.close()
