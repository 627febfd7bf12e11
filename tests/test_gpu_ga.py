"""GPU parity of the GA evolution loop (fcpp_ga_evolve), through the C ABI: the same seed gives the oracle's run
bit for bit -- final population, best route, histories, stats (routes are integers, fitness sums run in the reference's order)."""
import numpy as np
import pytest

import oracle as orc
from field_coverage_path_planning_amd import _lib as L
from field_coverage_path_planning_amd import engine as E
from field_coverage_path_planning_amd.genetic_algorithm_solver import GAConfig, GeneticAlgorithmSolver

pytestmark = pytest.mark.gpu


def _problem(seed, n, pop, asym=False):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0, 1000, size=(n, 2))
    D = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
    if asym:
        D = D + rng.uniform(0, 5, size=D.shape)
    routes = np.array([rng.permutation(n) for _ in range(pop)], dtype=np.int32)
    return D, routes


@pytest.mark.parametrize('n,pop,cfg,seed', [
    (12, 10, dict(max_generations=40, elite_size=2, tournament_size=3, convergence_threshold=50), 1),
    (129, 64, dict(max_generations=70, elite_size=20, tournament_size=5, convergence_threshold=50), 2 ** 40 + 17),
    (64, 200, dict(max_generations=120, elite_size=20, tournament_size=5, convergence_threshold=15, mutation_rate=0.3), 3),   # converges early
    (200, 128, dict(max_generations=33, elite_size=0, tournament_size=64, convergence_threshold=50, crossover_rate=0.5), 4),
    (2, 4, dict(max_generations=5, elite_size=1, tournament_size=1, convergence_threshold=50), 5),
    # tours too long for the one-launch-per-generation kernel's LDS: the two-stream path (pairs beside the bookkeeping)
    (700, 96, dict(max_generations=9, elite_size=10, tournament_size=5, convergence_threshold=50), 6),
    (520, 64, dict(max_generations=40, elite_size=4, tournament_size=3, convergence_threshold=6, mutation_rate=0.0, crossover_rate=0.0), 7),   # converges
    # a population beyond the LDS cache of the bookkeeping workgroup, with more than 64 elites (its general path)
    (40, 6400, dict(max_generations=4, elite_size=70, tournament_size=4, convergence_threshold=50), 8),
])
def test_evolution_equals_oracle(n, pop, cfg, seed):
    D, routes = _problem(seed % 1000, n, pop, asym=(n == 64))
    c = GAConfig(population_size=pop, **cfg)
    final, best, hb, ha, res = E.ga_evolve(D, routes, c, seed=seed)
    ofinal, obest, ohb, oha, ores = orc.ga_evolve(D, routes, population_size=pop, seed=seed,
                                                  **{k: getattr(c, k) for k in ('max_generations', 'crossover_rate', 'mutation_rate',
                                                                               'elite_size', 'tournament_size', 'convergence_threshold')})
    assert (res.generations, res.convergence_gen) == (ores.generations, ores.convergence_gen)
    assert res.best_distance == ores.best_distance and res.best_fitness == ores.best_fitness
    assert np.array_equal(final.cpu().numpy(), ofinal) and np.array_equal(best.cpu().numpy(), obest)
    assert np.array_equal(hb, ohb) and np.array_equal(ha, oha)
    assert all(sorted(r) == list(range(n)) for r in final.cpu().numpy())


def test_bad_configurations_are_refused():
    D, routes = _problem(1, 10, 9)
    with pytest.raises(L.FcppError):
        E.ga_evolve(D, routes, GAConfig(population_size=9))            # odd population
    D, routes = _problem(1, 10, 8)
    with pytest.raises(L.FcppError):
        E.ga_evolve(D, routes, GAConfig(population_size=8, elite_size=8))
    with pytest.raises(L.FcppError):
        E.ga_evolve(D, routes, GAConfig(population_size=8, elite_size=2, tournament_size=9))


def test_solver_solve_on_device():
    """GeneticAlgorithmSolver(config).solve(D) as the reference's callers use it (multi_vehicle_planner.py): route from the depot,
    stats keys, histories; a seed reproduces the run; the tour is much shorter than a random one."""
    D, _ = _problem(8, 40, 2)
    cfg = GAConfig(population_size=200, max_generations=150)
    s1, s2 = GeneticAlgorithmSolver(cfg, seed=11), GeneticAlgorithmSolver(cfg, seed=11)
    r1, st1 = s1.solve(D, verbose=False)
    r2, st2 = s2.solve(D, verbose=False)
    assert r1 == r2 and st1 == st2
    assert r1[0] == 0 and sorted(r1) == list(range(40))
    assert set(st1) == {'generations', 'best_distance', 'best_fitness', 'convergence_gen'}
    assert len(s1.best_fitness_history) == len(s1.avg_fitness_history) == st1['generations']
    assert abs(s1._calculate_distance(r1, D) - st1['best_distance']) < 1e-9 * st1['best_distance']
    rnd = np.mean([s1._calculate_distance(list(np.random.default_rng(k).permutation(40)), D) for k in range(5)])
    assert st1['best_distance'] < 0.5 * rnd


def test_distance_matrix_and_connections_vs_oracle(golden_ga):
    """Scheduler inputs: the matrix the GA consumes (MVP:229-259) and the exit -> entry connection search (MFP:290-320)."""
    nodes = golden_ga['dm_nodes']
    D = E.distance_matrix(nodes).cpu().numpy()
    assert np.array_equal(D, orc.distance_matrix(nodes))                       # same arithmetic: bit-equal with the oracle
    np.testing.assert_allclose(D, golden_ga['dm_D'], rtol=4.5e-16, atol=0)     # 1 ulp of the reference (numpy's dot kernel)
    # fed straight into the GA fitness: bit-exact tour lengths against the oracle on the same matrix
    rng = np.random.default_rng(3)
    routes = np.array([rng.permutation(len(nodes)) for _ in range(32)], dtype=np.int32)
    dist, _ = E.ga_fitness(routes, D)
    assert np.array_equal(dist.cpu().numpy(), orc.ga_distance(routes, D))
    # connections: quadrilateral corners as exit / entry candidates (MFP:136-140), a depot with one candidate, an empty list, ties
    fl, tl = [], []
    for p in range(200):
        a = rng.uniform(-500, 500, size=(4 if p % 7 else 1, 2))
        b = rng.uniform(-500, 500, size=(4 if p % 5 else 1, 2))
        if p % 11 == 0:
            b = np.vstack([b, b[::-1]])              # every distance twice: the first pair must win
        if p == 13:
            a = np.zeros((0, 2))
        fl.append(a); tl.append(b)
    bf, bt, bd = E.best_connections(fl, tl)
    for p in range(200):
        want = orc.best_connection(fl[p], tl[p])
        assert (int(bf[p]), int(bt[p])) == want[:2] and (bd[p] == want[2]), p
    big_f, big_t = rng.uniform(0, 1, size=(300, 2)), rng.uniform(0, 1, size=(257, 2))      # more products than lanes
    bf, bt, bd = E.best_connections([big_f], [big_t])
    assert (int(bf[0]), int(bt[0]), float(bd[0])) == orc.best_connection(big_f, big_t)


def test_scheduler_inputs_vs_the_reference_multi_field_planner(golden_mfp):
    """fcpp_distance_matrix / fcpp_best_connections against what the reference's own MultiFieldPlannerV38 computed
    (_calculate_distance_matrix MFP:263-288, _find_best_connection MFP:290-320; golden_mfp.npz), the tie included."""
    g = golden_mfp
    nodes = np.vstack([g['depot'][None, :], g['centroids']])
    D = E.distance_matrix(nodes).cpu().numpy()
    np.testing.assert_allclose(D, g['D'], rtol=4.5e-16, atol=0)
    cand = lambda node: g['depot'][None, :] if node == 0 else g['vertices'][node - 1]
    route = g['route']
    fl = [cand(route[k]) for k in range(len(route) - 1)]
    tl = [cand(route[k + 1]) for k in range(len(route) - 1)]
    bf, bt, bd = E.best_connections(fl, tl)
    for k in range(len(fl)):
        assert np.array_equal(fl[k][bf[k]], g['conn_from'][k]) and np.array_equal(tl[k][bt[k]], g['conn_to'][k]), k
    np.testing.assert_allclose(bd, g['conn_dist'], rtol=4.5e-16, atol=0)
