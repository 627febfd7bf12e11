"""One rank of tests/test_gpu_sharding.py::test_ga_population_sharded_on_the_gpu: sharding.ga_fitness_sharded with the REAL
fcpp_ga_fitness on cuda:0, process group over gloo (the ranks share the one GPU of the test box, so the all-gather carries host
tensors; the block cut, the per-block evaluation and the gather are the code a multi-GPU job runs).
usage: _shard_ga_gpu_worker.py RANK WORLD PORT POP N OUT_PREFIX"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rank, world, port, pop, n, out = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
                                  sys.argv[6])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

from field_coverage_path_planning_amd import sharding as S, workloads as WL      # noqa: E402

dist.init_process_group('gloo', rank=rank, world_size=world)
D, routes = WL.cfg4_ga(n, pop)
fit, dst = S.ga_fitness_sharded(torch.from_numpy(routes), torch.from_numpy(D), device=0, with_distance=True)
assert fit.is_cuda and fit.shape == (pop,)
np.savez(f'{out}.{rank}.npz', fit=fit.cpu().numpy(), dist=dst.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
