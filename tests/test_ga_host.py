"""Host-side surface of the GA mirror (CPU; the kernels themselves are covered by the -m gpu tests).  The evolution operators
have no host implementation in the product: what is checked here is the oracle's order crossover against a plain restatement
of GA:214-242, and the mirror's configuration record / initial population."""
import numpy as np

import oracle as orc
from field_coverage_path_planning_amd.genetic_algorithm_solver import GAConfig, GeneticAlgorithmSolver


def _ox_reference_semantics(p1, p2, a, b):
    """what GA:214-242 does for one child (child keeps p1[a:b], filled from p2 starting after b)"""
    n = len(p1)
    child = [None] * n
    child[a:b] = list(p1[a:b])
    pos = b
    for gene in list(p2[b:]) + list(p2[:b]):
        if gene not in child:
            if pos >= n:
                pos = 0
            child[pos] = gene
            pos += 1
    return child


def test_oracle_order_crossover_matches_reference_semantics():
    rng = np.random.default_rng(0)
    for n in (5, 10, 33, 128):
        for _ in range(40):
            p1, p2 = rng.permutation(n).astype(np.int32), rng.permutation(n).astype(np.int32)
            a, b = sorted(int(v) for v in rng.choice(n, size=2, replace=False))
            c1, c2 = orc.ga_ox(p1, p2, a, b)
            assert c1.tolist() == _ox_reference_semantics(p1, p2, a, b)
            assert c2.tolist() == _ox_reference_semantics(p2, p1, a, b)


def test_config_defaults_and_initial_population():
    assert GAConfig() == GAConfig(200, 500, 0.85, 0.02, 20, 5, 50)                 # GA:20-29
    s = GeneticAlgorithmSolver(GAConfig(population_size=64, elite_size=4), seed=1)
    pop = s._initialize_population(37)
    assert pop.shape == (64, 37) and pop.dtype == np.int32
    assert all(sorted(r.tolist()) == list(range(37)) for r in pop)
    assert [int(r[0]) for r in pop[32:]] == [i % 37 for i in range(32)]            # GA:160-164
    for name in ('_selection', '_crossover', '_mutation', '_elitism'):             # device-only (fcpp_ga_evolve)
        assert not hasattr(s, name)
