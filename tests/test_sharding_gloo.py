"""The N>1 path on CPU: world_size-2 gloo job exercising field_coverage_path_planning_amd/sharding.py
(partition on analytic point counts + final gather of per-field stats).  The per-rank compute is replaced by the
CPU oracle (test infrastructure); on a GPU box the same code path runs engine.Batch over RCCL."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _specs(n, seed=65536):
    from field_coverage_path_planning_amd import engine as E
    rng = np.random.default_rng(seed)
    LH = rng.uniform(60, 700, size=(n, 2))
    return [E.FieldSpec(field_length=float(a), field_width=float(b),
                        start_point=(float(a) * 0.9, float(b) * 0.1) if i % 3 == 0 else None)
            for i, (a, b) in enumerate(LH)]


def _oracle_compute(specs, vehicle, options):
    """(n, 13) int64 tensor laid out like fcpp_field_stats, computed by the oracle."""
    import oracle as orc
    from field_coverage_path_planning_amd import _lib as L
    rows = np.zeros((len(specs), L.STATS_WORDS), dtype=np.int64)
    for i, s in enumerate(specs):
        rc, p = orc.plan_field(orc.make_field(L=s.field_length, H=s.field_width, start=s.start_point),
                               orc.Vehicle.make(), orc.Options.make(options.turn_model, options.clothoid_fit,
                                                                    options.sample_spacing, options.clothoid_frac,
                                                                    options.geofence_tol))
        assert rc == 0
        d = np.array([p.main_len_m, p.main_time_pre_s, p.main_time_s, p.head_len_m, p.head_time_pre_s, p.head_time_s,
                      p.max_kappa, p.max_alat, p.max_jump], dtype=np.float64)
        rows[i, :9] = d.view(np.int64)
        rows[i, 9:] = [p.n_viol, p.n_outside, p.n_in_obstacle, p.n_adjusted]
    return torch.from_numpy(rows)


def _oracle_compute_with_points(specs, vehicle, options):
    """stats as above plus the block's point arrays (x, y, kappa, v, flagseg), as engine.Batch would hand them over"""
    import oracle as orc
    stats = _oracle_compute(specs, vehicle, options)
    xs, ys, ks, vs, fs = [], [], [], [], []
    for s in specs:
        _, p = orc.plan_field(orc.make_field(L=s.field_length, H=s.field_width, start=s.start_point), orc.Vehicle.make(),
                              orc.Options.make(options.turn_model, options.clothoid_fit, options.sample_spacing,
                                               options.clothoid_frac, options.geofence_tol))
        xs.append(p.xy[:, 0]); ys.append(p.xy[:, 1]); ks.append(p.kappa); vs.append(p.v); fs.append(p.flagseg.view(np.int32))
    cat = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.concatenate(a) if a else np.zeros(0), dtype=dt))
    return stats, [cat(xs, np.float64), cat(ys, np.float64), cat(ks, np.float64), cat(vs, np.float64), cat(fs, np.int32)]


def _worker_points(rank, world_size, port, n_fields, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    from field_coverage_path_planning_amd import engine as E
    from field_coverage_path_planning_amd import sharding as S
    res = S.plan_sharded(_specs(n_fields), E.make_vehicle(), E.make_options(), compute=_oracle_compute_with_points,
                         gather_points=True)
    if rank == 0:
        total = sum(i.n_main + i.n_head for i in res.infos)
        assert len(res.points_all) == 5 and all(a.numel() == total for a in res.points_all)
        np.savez(out_path, stats=res.stats_all.numpy(), **{f'a{k}': a.numpy() for k, a in enumerate(res.points_all)})
    else:
        assert res.points_all is None and res.stats_all is None
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world_size, port, n_fields, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    from field_coverage_path_planning_amd import engine as E
    from field_coverage_path_planning_amd import sharding as S
    specs = _specs(n_fields)
    res = S.plan_sharded(specs, E.make_vehicle(), E.make_options(), compute=_oracle_compute)
    lo, hi = res.block
    blocks = S.partition_by_points([i.n_main + i.n_head for i in res.infos], world_size)
    assert blocks[rank] == (lo, hi)
    if rank == 0:
        assert res.stats_all.shape == (n_fields, 13)
        np.save(out_path, res.stats_all.numpy())
    else:
        assert res.stats_all is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world_size', [2, 3])
def test_sharded_stats_equal_single_process(tmp_path, world_size):
    n_fields = 11
    out = str(tmp_path / 'stats.npy')
    port = 29500 + (os.getpid() % 2000) + world_size
    mp.spawn(_worker, args=(world_size, port, n_fields, out), nprocs=world_size, join=True)
    sharded = np.load(out)
    from field_coverage_path_planning_amd import engine as E
    single = _oracle_compute(_specs(n_fields), E.make_vehicle(), E.make_options()).numpy()
    assert np.array_equal(sharded, single)          # byte-identical for any shard count


@pytest.mark.parametrize('world_size', [2, 4, 8])
def test_sharded_point_arrays_equal_single_process(tmp_path, world_size):
    """The optional gather of the point arrays (direct sends into the root's slices, dist.batch_isend_irecv): the root's arrays are
    byte-identical to one process planning every field; world_size 4 over 5 fields leaves blocks of one field (and possibly none);
    world_size 8 over 21 fields: the cut and both gathers of a whole node's ranks (the eight-way job itself needs an eight-GPU node)."""
    n_fields = 5 if world_size == 4 else (21 if world_size == 8 else 9)
    out = str(tmp_path / 'points.npz')
    port = 31500 + (os.getpid() % 2000) + world_size
    mp.spawn(_worker_points, args=(world_size, port, n_fields, out), nprocs=world_size, join=True)
    got = np.load(out)
    from field_coverage_path_planning_amd import engine as E
    stats, arrays = _oracle_compute_with_points(_specs(n_fields), E.make_vehicle(), E.make_options())
    assert np.array_equal(got['stats'], stats.numpy())
    for k, a in enumerate(arrays):
        assert np.array_equal(got[f'a{k}'], a.numpy()), k


def test_partition_by_points_properties():
    from field_coverage_path_planning_amd.sharding import partition_by_points
    rng = np.random.default_rng(3)
    for trial in range(50):
        n = int(rng.integers(0, 40))
        counts = rng.integers(100, 100000, size=n)
        for ws in (1, 2, 3, 4, 8):
            blocks = partition_by_points(counts, ws)
            assert len(blocks) == ws and blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(ws - 1))      # contiguous, in rank order
            assert all(lo <= hi for lo, hi in blocks)
            if n >= 4 * ws:
                loads = [int(counts[lo:hi].sum()) for lo, hi in blocks]
                assert max(loads) <= counts.sum() / ws + counts.max()                 # balanced to within one field
    assert partition_by_points([], 4) == [(0, 0)] * 4
    assert partition_by_points([5, 5, 5, 5], 2) == [(0, 2), (2, 4)]


def test_plan_count_feeds_the_partition():
    """The partition is computed from fcpp_plan_count alone (host-only, no GPU)."""
    from field_coverage_path_planning_amd import engine as E
    from field_coverage_path_planning_amd.sharding import partition_by_points
    infos = E.plan_count(_specs(64), E.make_vehicle(), E.make_options(1, 0.25))
    counts = [i.n_main + i.n_head for i in infos]
    blocks = partition_by_points(counts, 8)
    loads = [sum(counts[lo:hi]) for lo, hi in blocks]
    assert max(loads) / (sum(counts) / 8) < 1.25


def _oracle_ga_compute(routes_block, D, order_mode):
    import oracle as orc
    r = routes_block.numpy().astype(np.int32)
    Dn = D.numpy() if isinstance(D, torch.Tensor) else np.asarray(D)
    if len(r) == 0:
        z = torch.zeros(0, dtype=torch.float64)
        return z, z
    return torch.from_numpy(orc.ga_distance(r, Dn)), torch.from_numpy(orc.ga_fitness(r, Dn))


def _ga_case(pop, n=17, seed=4096):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0, 1000, size=(n, 2))
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    routes = np.stack([rng.permutation(n) for _ in range(pop)]).astype(np.int32)
    return torch.from_numpy(D), torch.from_numpy(routes)


def _worker_ga(rank, world_size, port, pop, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    from field_coverage_path_planning_amd import sharding as S
    D, routes = _ga_case(pop)
    fit, dst = S.ga_fitness_sharded(routes, D, compute=_oracle_ga_compute, with_distance=True)
    only = S.ga_fitness_sharded(routes, D, compute=_oracle_ga_compute)
    assert fit.shape == (pop,) and torch.equal(only, fit)           # every rank holds the whole population's fitness
    np.savez(out_path + f'.{rank}.npz', fit=fit.numpy(), dist=dst.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world_size,pop', [(2, 37), (3, 64), (4, 3)])
def test_ga_population_sharded_equals_single_process(tmp_path, world_size, pop):
    """SURVEY.md 8e, GA: the population is cut into contiguous blocks, each rank evaluates its own, one all-gather of 8 B per
    chromosome (16 with the tour lengths) leaves the whole population's fitness on EVERY rank, byte-identical to one process; block
    sizes that differ by one (37 over 2, 64 over 3) and empty blocks (3 over 4)."""
    out = str(tmp_path / 'ga')
    port = 27500 + (os.getpid() % 2000) + world_size
    mp.spawn(_worker_ga, args=(world_size, port, pop, out), nprocs=world_size, join=True)
    D, routes = _ga_case(pop)
    d_ref, f_ref = _oracle_ga_compute(routes, D, 0)
    for rank in range(world_size):
        got = np.load(out + f'.{rank}.npz')
        assert np.array_equal(got['fit'], f_ref.numpy()) and np.array_equal(got['dist'], d_ref.numpy()), rank


def test_partition_even_properties():
    from field_coverage_path_planning_amd.sharding import partition_even
    for n in (0, 1, 3, 37, 4096):
        for ws in (1, 2, 3, 4, 8):
            b = partition_even(n, ws)
            assert len(b) == ws and b[0][0] == 0 and b[-1][1] == n and all(b[k][1] == b[k + 1][0] for k in range(ws - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
