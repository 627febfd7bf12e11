"""Mirror of the reference's `genetic_algorithm_solver` surface (GAConfig, GeneticAlgorithmSolver) with the tour-length
fitness on the GPU.

Hot path: `_calculate_distance` / `_calculate_fitness` (GA:168-181) -> `fcpp_ga_fitness`, one
wavefront per chromosome, left-to-right float64 summation (bit-exact with the reference) for the WHOLE population per call.
The evolution loop around it (selection, order crossover, swap mutation, elitism; GA:183-268) is host-side control and not
part of the accelerated path; it is restated here with vectorised numpy operators so that `solve()` works as in the
reference.  The reference draws from the unseeded stdlib `random`, so evolved routes are not reproducible there either;
this class takes an optional `seed`.
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

    # ---- host-side evolution loop (GA:44-135) ------------------------------------------------------------------
    def solve(self, distance_matrix: np.ndarray, verbose: bool = True) -> Tuple[List[int], dict]:
        cfg, rng = self.config, self._rng
        D = np.ascontiguousarray(distance_matrix, dtype=np.float64)
        n = len(D)
        if verbose:
            print(f"\n[遗传算法] 开始优化...  节点数: {n}  种群大小: {cfg.population_size}  最大代数: {cfg.max_generations}")
        half = cfg.population_size // 2
        # GA:137-166: both halves are random permutations (the "greedy" half only fixes its first node)
        pop = np.array([rng.permutation(n) for _ in range(2 * half)], dtype=np.int32)
        for i in range(half):
            row = pop[half + i]
            j = int(np.where(row == i % n)[0][0])
            row[0], row[j] = row[j], row[0]
        dist, fit = self.evaluate_population(pop, D)
        best_i = int(np.argmax(fit))
        best_route, best_fit, best_dist = pop[best_i].copy(), float(fit[best_i]), float(dist[best_i])
        stall, generation = 0, -1
        for generation in range(cfg.max_generations):
            sel = self._selection(pop, fit)
            off = self._crossover(sel)
            off = self._mutation(off)
            pop = self._elitism(pop, off, fit)
            dist, fit = self.evaluate_population(pop, D)
            i = int(np.argmax(fit))
            if fit[i] > best_fit:
                best_fit, best_route, best_dist, stall = float(fit[i]), pop[i].copy(), float(dist[i]), 0
                if verbose and generation % 50 == 0:
                    print(f"  第 {generation} 代: 最优距离 = {best_dist:.1f}m")
            else:
                stall += 1
            self.best_fitness_history.append(best_fit)
            self.avg_fitness_history.append(float(np.mean(fit)))
            if stall >= cfg.convergence_threshold:
                if verbose:
                    print(f"  第 {generation} 代: 收敛 (连续 {cfg.convergence_threshold} 代无改进)")
                break
        route = [int(g) for g in best_route]
        k = route.index(0)                         # GA:118-120: start from the depot
        final = route[k:] + route[:k]
        stats = {'generations': generation + 1, 'best_distance': best_dist, 'best_fitness': best_fit,
                 'convergence_gen': generation - stall}
        if verbose:
            print(f"[遗传算法] 优化完成! 总代数: {stats['generations']}  最优距离: {best_dist:.1f}m")
        return final, stats

    # GA:183-196 tournament selection
    def _selection(self, pop, fit):
        m = len(pop)
        k = min(self.config.tournament_size, m)
        # `k` distinct contestants per slot: rank random keys
        cand = np.argsort(self._rng.random((m, m)), axis=1)[:, :k] if m <= 512 else \
            np.stack([self._rng.choice(m, size=k, replace=False) for _ in range(m)])
        win = cand[np.arange(m), np.argmax(fit[cand], axis=1)]
        return pop[win].copy()

    # GA:198-242 order crossover (OX) on consecutive pairs
    def _crossover(self, sel):
        m, n = sel.shape
        p1 = sel[0::2]
        p2 = sel[1::2] if m % 2 == 0 else np.vstack([sel[1::2], sel[:1]])
        pairs = len(p1)
        cuts = np.sort(np.argsort(self._rng.random((pairs, n)), axis=1)[:, :2], axis=1)   # two distinct cut points
        do = self._rng.random(pairs) < self.config.crossover_rate
        c1 = self._ox(p1, p2, cuts)
        c2 = self._ox(p2, p1, cuts)
        c1[~do], c2[~do] = p1[~do], p2[~do]
        out = np.empty((2 * pairs, n), dtype=sel.dtype)
        out[0::2], out[1::2] = c1, c2
        return out

    @staticmethod
    def _ox(keep, fill, cuts):
        """child[a:b] = keep[a:b]; the other positions, starting at b and wrapping, take fill's genes in the order they
        appear from position b on, skipping genes already present (GA:225-237)."""
        pairs, n = keep.shape
        a, b = cuts[:, :1], cuts[:, 1:]
        pos = np.arange(n)[None, :]
        in_seg = (pos >= a) & (pos < b)
        present = np.zeros((pairs, n), dtype=bool)
        rows = np.repeat(np.arange(pairs), n).reshape(pairs, n)
        present[rows[in_seg], keep[in_seg]] = True
        order = (pos + b) % n                                  # positions b, b+1, ..., wrapping
        donor = np.take_along_axis(fill, order, axis=1)        # fill's genes from b on
        free = ~np.take_along_axis(present, donor, axis=1)     # ... that the child does not have yet
        # stable partition: free genes first, in order
        idx = np.argsort(~free, axis=1, kind='stable')
        donor_sorted = np.take_along_axis(donor, idx, axis=1)
        child = keep.copy()
        n_free = n - (b - a)                                   # per pair
        slot = np.arange(n)[None, :]
        use = slot < n_free
        tgt = np.take_along_axis(order, slot % n, axis=1)      # target positions b, b+1, ... (the segment comes last)
        child[rows[use], tgt[use]] = donor_sorted[use]
        return child

    # GA:244-252 swap mutation
    def _mutation(self, pop):
        m, n = pop.shape
        hit = np.where(self._rng.random(m) < self.config.mutation_rate)[0]
        for r in hit:
            i, j = self._rng.choice(n, size=2, replace=False)
            pop[r, i], pop[r, j] = pop[r, j], pop[r, i]
        return pop

    # GA:254-268 elitism
    def _elitism(self, old_pop, new_pop, old_fit):
        e = self.config.elite_size
        elite = np.argsort(old_fit)[-e:]
        return np.vstack([new_pop[:-e], old_pop[elite]]) if e > 0 else new_pop
