"""One rank of tests/test_gpu_sharding.py: sharding.plan_sharded with the REAL engine.Batch on cuda:0, process group over gloo
(two ranks share the one GPU of the test box; RCCL refuses two ranks on one device, so the collective legs carry host tensors --
the partition, the batch per block, the stats gather and the point-array gather are the code a multi-GPU job runs).
usage: _shard_gpu_worker.py RANK WORLD PORT N_FIELDS OUT_NPZ"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rank, world, port, n_fields, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

from field_coverage_path_planning_amd import engine as E, sharding as S, workloads as WL      # noqa: E402

dist.init_process_group('gloo', rank=rank, world_size=world)
V = WL.cfg5_parallelograms(n_fields, seed=65536)
specs = WL.specs_from_vertices(E, V)
res = S.plan_sharded(specs, E.make_vehicle(), E.make_options(), device=0, gather_points=True)
if rank == 0:
    np.savez(out, stats=res.stats_all.numpy(), blocks=np.array(res.blocks), **{f'a{k}': a.numpy() for k, a in enumerate(res.points_all)})
else:
    assert res.stats_all is None and res.points_all is None
dist.barrier()
dist.destroy_process_group()
