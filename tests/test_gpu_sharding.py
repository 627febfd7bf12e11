"""The N > 1 path with the real engine.Batch: a world-size-2 (and 3) job on the one GPU of the test box, process group over gloo
(tests/_shard_gpu_worker.py).  The root's gathered stats and point arrays must be byte-identical to one process planning every field.
The workers are started before this process initialises the GPU-side comparison run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('world', [2, 3])
def test_real_batch_sharded_equals_single_process(tmp_path, world):
    n_fields = 96
    out = str(tmp_path / 'sharded.npz')
    port = 33500 + (os.getpid() % 2000) + world
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', '_shard_gpu_worker.py'), str(r), str(world), str(port),
                               str(n_fields), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode(errors='replace') for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)
    got = np.load(out)

    from field_coverage_path_planning_amd import engine as E, workloads as WL
    specs = WL.specs_from_vertices(E, WL.cfg5_parallelograms(n_fields, seed=65536))
    batch = E.Batch(specs, E.make_vehicle(), E.make_options())
    res = batch.run()
    assert np.array_equal(got['stats'], res.stats_raw.cpu().numpy())              # byte-identical for any shard count
    for k, a in enumerate((res.x, res.y, res.kappa, res.v, res.flagseg)):
        assert np.array_equal(got[f'a{k}'], a.cpu().numpy()), k
    blocks = got['blocks']
    assert blocks[0, 0] == 0 and blocks[-1, 1] == n_fields and (blocks[1:, 0] == blocks[:-1, 1]).all()
    loads = [sum(i.n_main + i.n_head for i in batch.info[lo:hi]) for lo, hi in blocks]
    assert max(loads) <= batch.total_points / world + max(i.n_main + i.n_head for i in batch.info)
    batch.close()


def test_stats_and_points_identical_for_1_2_4_8_shards():
    """The cut of sharding.partition_by_points for 1, 2, 4 and 8 ranks, every block planned as a batch of its own (what each rank of
    an N-GPU job does), blocks concatenated in rank order: statistics and point arrays byte-identical to the single batch.  Mixed
    workload: parallelograms at the reference sampling and dense clothoid rectangles, so that every kernel of the fused pipeline
    takes part."""
    import torch
    from field_coverage_path_planning_amd import engine as E, sharding as S, workloads as WL
    veh = E.make_vehicle()
    for specs, opt in ((WL.specs_from_vertices(E, WL.cfg5_parallelograms(160, seed=7)), E.make_options()),
                       (WL.specs_from_lh(E, WL.cfg2_rectangles(40, seed=9)), E.make_options(1, 0.5))):
        infos = E.plan_count(specs, veh, opt)
        counts = [i.n_main + i.n_head for i in infos]
        # the sizing a sharded job uses: on the device for the reference's sampling, on the host for the dense batch -- the same counts
        assert np.array_equal(E.plan_points(specs, veh, opt), np.asarray(counts, dtype=np.int64))
        whole = E.Batch(specs, veh, opt)
        ref = whole.run()
        ref_stats = ref.stats_raw.cpu().numpy()
        ref_arrays = [a.cpu().numpy() for a in (ref.x, ref.y, ref.kappa, ref.v, ref.flagseg)]
        for world in (1, 2, 4, 8):
            blocks = S.partition_by_points(counts, world)
            assert len(blocks) == world and blocks[0][0] == 0 and blocks[-1][1] == len(specs)
            stats, arrays = [], [[] for _ in range(5)]
            for lo, hi in blocks:
                if hi == lo:
                    continue
                b = E.Batch(specs[lo:hi], veh, opt)
                r = b.run()
                stats.append(r.stats_raw.cpu().numpy())
                for k, a in enumerate((r.x, r.y, r.kappa, r.v, r.flagseg)):
                    arrays[k].append(a.cpu().numpy())
                b.close()
            assert np.array_equal(np.concatenate(stats), ref_stats), world
            for k in range(5):
                assert np.array_equal(np.concatenate(arrays[k]), ref_arrays[k]), (world, k)
        whole.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize('world,pop,n', [(2, 4096, 128), (3, 1001, 129)])
def test_ga_population_sharded_on_the_gpu(tmp_path, world, pop, n):
    """SURVEY.md 8e, GA: the population cut into contiguous blocks over the ranks, fcpp_ga_fitness per block, one all-gather of the
    fitness (and tour lengths): every rank ends with the whole population's values, byte-identical to one process and to the oracle
    (cfg4's size; and blocks of unequal length with tours longer than two wavefront passes)."""
    out = str(tmp_path / 'ga')
    port = 35500 + (os.getpid() % 2000) + world
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', '_shard_ga_gpu_worker.py'), str(r), str(world), str(port),
                               str(pop), str(n), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode(errors='replace') for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)

    import oracle as orc
    from field_coverage_path_planning_amd import engine as E, workloads as WL
    D, routes = WL.cfg4_ga(n, pop)
    dist1, fit1 = E.ga_fitness(routes, D)
    f_orc, d_orc = orc.ga_fitness(routes, D), orc.ga_distance(routes, D)
    assert np.array_equal(fit1.cpu().numpy(), f_orc) and np.array_equal(dist1.cpu().numpy(), d_orc)
    for r in range(world):
        got = np.load(f'{out}.{r}.npz')
        assert np.array_equal(got['fit'], f_orc) and np.array_equal(got['dist'], d_orc), r
