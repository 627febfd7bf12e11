"""Mirror of the reference's `genetic_algorithm_solver` surface (GAConfig, GeneticAlgorithmSolver) with the tour-length
fitness on the GPU.

Hot path: `_calculate_distance` / `_calculate_fitness` (GA:168-181) -> `fcpp_ga_fitness`, one
wavefront per chromosome, left-to-right float64 summation (bit-exact with the reference) for the WHOLE population per call.
The evolution loop around it (selection, order crossover, swap mutation, elitism, best tracking, convergence; GA:64-115,
183-268) runs on the GPU as well (`fcpp_ga_evolve`: two launches per generation, no host round trip), its random decisions
drawn from the counter-based generator Philox4x32-10.  The reference draws from the unseeded stdlib `random`, so evolved
routes are not reproducible there; this class takes an optional `seed`, and a given seed reproduces a run bit for bit
(and equals the CPU oracle's run).  population_size must be even.
"""
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import engine as E

__all__ = ['GAConfig', 'GeneticAlgorithmSolver']


@dataclass
class GAConfig:
    """遗传算法配置参数 (GA:20-29)"""
    population_size: int = 200
    max_generations: int = 500
    crossover_rate: float = 0.85
    mutation_rate: float = 0.02
    elite_size: int = 20
    tournament_size: int = 5
    convergence_threshold: int = 50


class GeneticAlgorithmSolver:
    """遗传算法 TSP 求解器 (GA:32-268); fitness on the GPU."""

    def __init__(self, config: GAConfig = None, seed: int = None, device: int = None):
        self.config = config or GAConfig()
        self.best_fitness_history = []
        self.avg_fitness_history = []
        self._rng = np.random.default_rng(seed)
        self._device = device

    # ---- hot path ------------------------------------------------------------------------------------------
    def evaluate_population(self, population, distance_matrix, order_mode: int = 0):
        """-> (distances, fitness) numpy arrays for a (pop, n) array of tours: one kernel launch."""
        d, f = E.ga_fitness(np.asarray(population, dtype=np.int32), distance_matrix, order_mode=order_mode,
                            device=self._device)
        return d.cpu().numpy(), f.cpu().numpy()

    def _calculate_distance(self, route: List[int], distance_matrix: np.ndarray) -> float:
        """GA:174-181"""
        return float(self.evaluate_population([list(route)], distance_matrix)[0][0])

    def _calculate_fitness(self, route: List[int], distance_matrix: np.ndarray) -> float:
        """GA:168-172"""
        return float(self.evaluate_population([list(route)], distance_matrix)[1][0])

    # ---- solve (GA:44-135): the whole evolution loop runs on the device -----------------------------------------------
    def solve(self, distance_matrix: np.ndarray, verbose: bool = True) -> Tuple[List[int], dict]:
        cfg, rng = self.config, self._rng
        D = np.ascontiguousarray(distance_matrix, dtype=np.float64)
        n = len(D)
        if verbose:
            print(f"\n[遗传算法] 开始优化...  节点数: {n}  种群大小: {cfg.population_size}  最大代数: {cfg.max_generations}")
        pop = self._initialize_population(n)
        # every random decision of the loop is drawn on the device from Philox4x32-10 keyed by this seed (include/fcpp.h)
        seed = int(rng.integers(0, 2 ** 63))
        _, best, hb, ha, res = E.ga_evolve(D, pop, cfg, seed=seed, device=self._device)
        self.best_fitness_history.extend(float(v) for v in hb)           # GA:106-107
        self.avg_fitness_history.extend(float(v) for v in ha)
        route = [int(g) for g in best.cpu().numpy()]
        k = route.index(0)                         # GA:118-120: start from the depot
        final = route[k:] + route[:k]
        stats = {'generations': res.generations, 'best_distance': res.best_distance, 'best_fitness': res.best_fitness,
                 'convergence_gen': res.convergence_gen}
        if verbose:
            print(f"[遗传算法] 优化完成! 总代数: {stats['generations']}  最优距离: {res.best_distance:.1f}m  "
                  f"收敛代数: {stats['convergence_gen']}")
        return final, stats

    def _initialize_population(self, num_nodes: int) -> np.ndarray:
        """GA:137-166: both halves are random permutations (the "greedy" half only fixes its first node, i % n)."""
        half = self.config.population_size // 2
        pop = np.array([self._rng.permutation(num_nodes) for _ in range(2 * half)], dtype=np.int32)
        for i in range(half):
            row = pop[half + i]
            j = int(np.where(row == i % num_nodes)[0][0])
            row[0], row[j] = row[j], row[0]
        return pop
